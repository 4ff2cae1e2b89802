// Phased K loop for the k = 3 convolutions on 256-channel x 254-frame tiles (8 waves of 128 ch x 64 frames): the same
// implicit GEMM as conv_gemm2_kernel (conv_gemm2_impl.h), same operand layouts, same accumulation order -- results are
// bit-identical -- with a K loop built for overlap instead of one `vmcnt(0)` + __syncthreads per stage:
//
//  * A stage (one tap of one 64-channel chunk) is cut into FOUR phases, one per 16-wide K slice over the whole wave
//    tile: 4 + 2 fragment reads (nothing is read twice), 8 MFMAs on 8 independent accumulators, 24 fragment registers.
//  * The two wave groups (waves 0-3 / 4-7: one wave of each group per SIMD) run half a phase apart with ONE raw
//    s_barrier per phase: group 0 takes it between its reads and its MFMAs, group 1 before its reads, so between two
//    barriers group 0 runs [M(p-1) R(p)] while group 1 runs [R(p-1) M(p-1)] -- on every SIMD one wave is in its MFMAs
//    (s_setprio 1) while the other reads fragments and issues LDS-DMA.
//  * THREE weight buffers and counted waits: the weight tile of stage s+2 is issued during stage s and retired by a
//    `s_waitcnt vmcnt(N)` in stage s+1 that leaves the younger pieces in flight, so every LDS-DMA piece has more than
//    a whole stage to land.  (With two buffers and a drain per stage the stage time is set by the DMA round trip:
//    removing the MFMAs from that loop changed its time by 4 %.)
//
// LDS: 2 activation + 3 weight buffers of exactly 32 KiB = the CU's 160 KiB, which is why the activation tile is
// 256 rows = frames t0-1 .. t0+254 (254 output frames per tile + its own halo, like conv_gemm3_kernel): accumulator
// columns 254, 255 read two rows past the tile (inside the block's own LDS) and are dropped by the epilogue.
//
// Per stage (c, j), stage index s = 3c + j, weight buffer = j (3 | stages per chunk); pieces = 1 KiB, 4 + 4 per wave:
//   phase 1: (j = 1: A(c+1) piece 2)        phase 2: W(s+2) pieces 0, 1
//   phase 3: WAIT, W(s+2) piece 2 (j = 0: A(c+1) piece 0)        phase 4: W(s+2) piece 3 (j = 0: A piece 1; j = 1: A piece 3)
// The wait must retire W(s+1) (and at j = 2 the whole of A(c+1)).  Younger than W(s+1)'s last piece at that point:
//   j = 0: W(s+2) p0,p1 -> vmcnt(2)     j = 1: A1, A2, W p0,p1 -> vmcnt(4)     j = 2: A3, W p0,p1, A3 needed -> vmcnt(2)
// Every piece is issued unconditionally (rows outside [0, T) read the zero page), so the counts hold for every wave;
// the last chunk, which issues less, drains with vmcnt(0).
// Hazards, counted in barriers ("ticks"; group 1 runs a phase one tick after group 0):
//   RAW  the wait sits in phase 3 (group 1: tick 4s+3), the first read of the retired buffer in phase 1 of the next
//        stage (group 0: tick 4s+4): a barrier every wave has passed lies between them.
//   WAR  W(s+3) overwrites W(s): last read in phase 4 of stage s (group 1: tick 4s+4, retired by lgkmcnt(0) inside the
//        tick), first issued in phase 2 of stage s+1 (group 0: tick 4s+5).  A(c+1) overwrites A(c-1), last read in
//        phase 4 of stage (c-1, 2); its first piece is issued in phase 3 of stage (c, 0).
// Measured (MI355X, FFN conv_2 of the headline solve, 48 stages): K loop 64 us = 67 % MFMA utilisation inside the loop
// (conv_gemm2_kernel: ~74 us); the kernel's remaining 36 us are its HBM-bound epilogue.
#pragma once
#include <type_traits>
#include "conv_gemm2_impl.h"

#ifndef ST_PRIO_PHASED
#define ST_PRIO_PHASED 0      // 1: s_setprio 1 around the MFMA clusters (the round-4 form; without it the solve is 0.2-0.5 % faster per kernel family, paired: profiles/r05_ab_setprio.txt)
#endif

namespace st {

#define ST_RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

template <class P, int EPI, bool TWO_SRC>
__global__ __launch_bounds__(512, 1)
void conv_gemm_phased3_kernel(const ConvGemmArgs g) {
    constexpr int BC = 256, BF = 256, BFV = BF - 2, WC = 2, WF = 4, TAPS = 3;
    using vec8 = typename P::vec8;
    constexpr int FC = 4, FF = 2, TC = BC / WC, TF = BF / WF;
    constexpr int A_BYTES = BF * 128, W_BYTES = BC * 128;
    constexpr int WPW = 4, APW = 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: 2 activation buffers (As = smem), then 3 weight buffers (Ws = smem + 2 * A_BYTES)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wave % WC, wf = wave / WC;
    const int grp = wave >> 2, ngrp = grp ^ 1;
    const int cin = g.c0 + g.c1 + g.c2;
    const int nch = cin >> 6;
    const int T = g.T;
    const unsigned char* zeros = (const unsigned char*)g.zeros;
    const int prow = lane >> 3;
    const int wrow = wc * TC + l31, arow0 = wf * TF + l31;

    const int total = g.n_items * g.tiles_f * g.tiles_c;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int tc = lin % g.tiles_c;
    const int rest = lin / g.tiles_c;
    const int tf = rest % g.tiles_f;
    const int n = rest / g.tiles_f;
    const int cbase = tc * BC, t0 = tf * BFV;
    if (g.t_lim && t0 >= g.t_lim[n % g.t_lim_mod]) return;      // ragged batch: this tile lies past the item's last needed frame

    // Accumulators.  The bias-initialised variants (EPI_ACT16 / EPI_GELU16: "start from the bias", g2_init_acc) keep the bias
    // as four 16-register tuples that enter as the C operand of the very first MFMAs (D = A B + bias: the same operation
    // as accumulating into a bias-initialised register, bit for bit) -- no 2 x 64 register copies next to the loop's
    // invariant address registers.
    constexpr bool BIAS_C = (EPI == EPI_ACT16 || EPI == EPI_GELU16);
    f32x16_t acc[FC][FF];
    f32x16_t bt[BIAS_C ? FC : 1];
    if constexpr (BIAS_C) {
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g.bias) bv = *(const float4*)(g.bias + cbase + wc * TC + a * 32 + 8 * q4 + 4 * hi);
                bt[a][4 * q4 + 0] = bv.x; bt[a][4 * q4 + 1] = bv.y; bt[a][4 * q4 + 2] = bv.z; bt[a][4 * q4 + 3] = bv.w;
            }
    } else {
        g2_init_acc<EPI, FC, FF>(acc, g, cbase + wc * TC, hi);
    }

    const unsigned char* a0 = (const unsigned char*)g.a0 + (size_t)(n % g.a0_mod) * T * g.c0 * 2;
    const unsigned char* a1 = TWO_SRC ? (const unsigned char*)g.a1 + (size_t)(n % g.a1_mod) * T * g.c1 * 2 : nullptr;
    const unsigned char* wsrc = (const unsigned char*)g.w + (size_t)n * g.w_item_stride;

    unsigned voffW[WPW], voffA0[APW], voffA1[TWO_SRC ? APW : 1];
    bool validA[APW];
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
        const int row = (wave * WPW + k) * 8 + prow;
        voffW[k] = (unsigned)((cbase + row) * TAPS * cin * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int k = 0; k < APW; ++k) {
        const int row = (wave * APW + k) * 8 + prow;
        const int t = t0 + row - 1;
        const unsigned segb = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        validA[k] = (t >= 0 && t < T);
        voffA0[k] = (unsigned)(t * g.c0 * 2) + segb;
        if constexpr (TWO_SRC) voffA1[k] = (unsigned)(t * g.c1 * 2) + segb;
    }
    // Everything wave-uniform in the loop lives in SGPRs and every per-lane address is computed ONCE, before the loop: on a
    // SIMD a VALU instruction is issued INSTEAD of an MFMA (rocprofv3 SQ counters, profiles/r03_attention_sq_counters_and_
    // ablation.txt), and the ~16 VALU per phase of per-phase address arithmetic cost a fifth of the loop.
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void_t*)smem);
    const unsigned ldsA = lds0, ldsW = lds0 + 2 * A_BYTES;
    const unsigned char* wsrc_s = sgpr_ptr64(wsrc);
    const unsigned char* a0_s = sgpr_ptr64(a0);
    const unsigned char* a1_s = TWO_SRC ? sgpr_ptr64(a1) : nullptr;
    auto issueW = [&](int c, int j, int buf, int k0, int k1) {
        const unsigned char* sb = wsrc_s + (size_t)(j * cin + (c << 6)) * 2;
#pragma unroll
        for (int k = 0; k < WPW; ++k)
            if (k >= k0 && k < k1) glds16o(sb, voffW[k], ldsW + buf * W_BYTES + (wave * WPW + k) * 1024);
    };
    auto issueA = [&](int c, int buf, int k0, int k1) {
        int ch0 = c << 6;
        if (ch0 >= g.c0 + g.c1) ch0 -= g.c0 + g.c1;
        if (!TWO_SRC || ch0 < g.c0) {
            const unsigned char* sb = a0_s + (size_t)ch0 * 2;
#pragma unroll
            for (int k = 0; k < APW; ++k)
                if (k >= k0 && k < k1) glds16bo(validA[k] ? sb + voffA0[k] : zeros, ldsA + buf * A_BYTES + (wave * APW + k) * 1024);
        } else if constexpr (TWO_SRC) {
            const unsigned char* sb = a1_s + (size_t)(ch0 - g.c0) * 2;
#pragma unroll
            for (int k = 0; k < APW; ++k)
                if (k >= k0 && k < k1) glds16bo(validA[k] ? sb + voffA1[k] : zeros, ldsA + buf * A_BYTES + (wave * APW + k) * 1024);
        }
    };


    // Fragment addresses: rows 32 apart share the swizzle term ((row >> 1) & 7), so the four channel fragments (and the
    // two frame fragments) of a wave are ONE per-lane address per k-slice plus a 4096-byte immediate.  The 4 weight addresses
    // (one per k-slice; a second set 64 KiB up for weight buffer 2, whose offset does not fit ds_read's 16-bit immediate) and
    // the 12 activation addresses (tap x k-slice: the swizzle term moves with the tap's row shift) are loop invariant: 20
    // registers, made opaque so that hipcc neither recomputes them per phase nor hoists anything else next to them.
    unsigned wadr[4], wadr2[4], aadr[3][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        wadr[ks] = ldsW + (unsigned)(wrow * 128 + (((ks * 2 + hi) ^ ((wrow >> 1) & 7)) << 4));
        wadr2[ks] = wadr[ks] + 2 * W_BYTES;
        asm volatile("" : "+v"(wadr[ks]), "+v"(wadr2[ks]));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int arow = arow0 + j;          // rows 256, 257 (last two frame columns at j = 2) fall into the next buffer
            aadr[j][ks] = ldsA + (unsigned)(arow * 128 + (((ks * 2 + hi) ^ ((arow >> 1) & 7)) << 4));
            asm volatile("" : "+v"(aadr[j][ks]));
        }
    }
    vec8 wfr[FC], afr[FF];
    auto load_frags = [&](int wbuf, unsigned abuf_off, int j, int ks) {      // wbuf, j, ks: compile-time after unrolling
        const unsigned wad = wbuf == 2 ? wadr2[ks] : wadr[ks];
        const unsigned woff = wbuf == 1 ? W_BYTES : 0;
        const unsigned aad = aadr[j][ks] + abuf_off;      // the one VALU instruction of a phase (chunk parity is a run-time value)
#pragma unroll
        for (int a = 0; a < FC; ++a) wfr[a] = as_vec8<P>(lds_read16(wad + woff + a * 4096));
#pragma unroll
        for (int b = 0; b < FF; ++b) afr[b] = as_vec8<P>(lds_read16(aad + b * 4096));
    };
    auto mma = [&](bool first) {      // first: compile-time after unrolling (the first phase of the first chunk)
        if constexpr (ST_PRIO_PHASED) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int b = 0; b < FF; ++b) {
                if constexpr (BIAS_C) acc[a][b] = P::mfma(wfr[a], afr[b], first ? bt[a] : acc[a][b]);
                else acc[a][b] = P::mfma(wfr[a], afr[b], acc[a][b]);
            }
        if constexpr (ST_PRIO_PHASED) __builtin_amdgcn_s_setprio(0);
    };
    // One phase = fragment reads (+ this phase's LDS-DMA issues), reads retired, 8 MFMAs, with ONE barrier: group 0 takes
    // it between its reads and its MFMAs, group 1 before its reads -- so between two barriers group 0 runs [M(p-1) R(p)]
    // while group 1 runs [R(p-1) M(p-1)]: on every SIMD one wave is in its MFMAs while the other reads / issues.
#define ST_BARRIER_IF(cond) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lnb_%=\n\ts_barrier\n.Lnb_%=:" :: "s"(cond) : "memory", "scc")
#define ST_PHASE(ks, ISSUE)                                      \
    __builtin_amdgcn_sched_barrier(0);                           \
    ST_BARRIER_IF(grp);                                          \
    load_frags(j, abuf_off, j, ks);                              \
    ISSUE;                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);                           \
    ST_BARRIER_IF(ngrp);                                         \
    mma(FIRST && j == 0 && ks == 0);                             \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: A(0), W(0) resident; W(1) in flight (retired by stage 0's wait)
    issueA(0, 0, 0, APW); issueW(0, 0, 0, 0, WPW);
    issueW(0, 1, 1, 0, WPW);
    ST_DMA_WAIT(4);
    __syncthreads();

    auto chunk = [&](int c, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const bool lastc = (c == nch - 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // stage s + 2 = (c, j + 2) or (c + 1, j - 1); its buffer is (j + 2) % 3
            const int c2 = (j == 0) ? c : c + 1, j2 = (j + 2) % 3;
            const bool w2 = (j == 0) || !lastc;
            const unsigned abuf_off = (unsigned)(c & 1) * A_BYTES;
            ST_PHASE(0, if (j == 1 && !lastc) issueA(c + 1, (c + 1) & 1, 2, 3))
            ST_PHASE(1, if (w2) issueW(c2, j2, j2, 0, 2))
            ST_PHASE(2, {
                if (lastc && j >= 1) ST_DMA_WAIT(0);
                else if (j == 1) ST_DMA_WAIT(4);
                else ST_DMA_WAIT(2);
                if (w2) issueW(c2, j2, j2, 2, 3);
                if (j == 0 && !lastc) issueA(c + 1, (c + 1) & 1, 0, 1);
            })
            ST_PHASE(3, {
                if (w2) issueW(c2, j2, j2, 3, 4);
                if (j == 0 && !lastc) issueA(c + 1, (c + 1) & 1, 1, 2);
                if (j == 1 && !lastc) issueA(c + 1, (c + 1) & 1, 3, 4);
            })
        }
    };
    // the first chunk is peeled when its first MFMAs take the bias as their C operand
    chunk(0, std::integral_constant<bool, BIAS_C>{});
    for (int c = 1; c < nch; ++c) chunk(c, std::false_type{});
#undef ST_PHASE
#undef ST_BARRIER_IF
    __syncthreads();

    if constexpr (EPI == EPI_ACT16 || EPI == EPI_GELU16)
        g2_epilogue_act16<P, BC, BF, WC, WF, EPI == EPI_GELU16>(acc, smem, g, n, t0, BFV, cbase, wave, lane);
    else if constexpr (EPI == EPI_SILU) g2_epilogue_silu<P, BC, BF, WC, WF>(acc, smem, g, n, t0, BFV, cbase, wave, lane);
    else g2_epilogue<P, EPI, BC, BF, WC, WF>(acc, (float*)smem, g, n, t0, BFV, cbase, wave, lane);
}

template <class P, int EPI>
static hipError_t launch_phased3(const ConvGemmArgs& a, hipStream_t s) {
    constexpr int BC = 256, BF = 256, LDS = 5 * 32768;
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_done_dev[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_phased3_kernel<P, EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_gemm_phased3_kernel<P, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_done_dev[dev_] = true;
    }
    if (!a.zeros || (a.cout % BC) != 0 || ((a.c0 | a.c1 | a.c2) & 63) != 0 || (a.c2 && a.c2 > a.c0) || a.ksplit > 1) return hipErrorInvalidValue;
    ConvGemmArgs b = a;
    b.tiles_f = (a.T + BF - 3) / (BF - 2);
    b.tiles_c = a.cout / BC;
    const int total = b.n_items * b.tiles_f * b.tiles_c;
    const int grid = ((total + 7) / 8) * 8;
    if (a.c1) hipLaunchKernelGGL((conv_gemm_phased3_kernel<P, EPI, true>), dim3(grid), dim3(512), LDS, s, b);
    else      hipLaunchKernelGGL((conv_gemm_phased3_kernel<P, EPI, false>), dim3(grid), dim3(512), LDS, s, b);
    return hipGetLastError();
}

}  // namespace st
