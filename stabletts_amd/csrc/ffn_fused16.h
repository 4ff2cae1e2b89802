// The fused FFN kernel of ffn_fused.h (same block / chunk / area / ring / phase structure, same LDS map, same LDS-DMA protocol --
// read that header first) with its two contractions on the 16x16x32 MFMA instead of 32x32x16.
//
// Why a second shape: the kernel is limited by the chip's POWER, not by issue slots (203 us on real data, 155 us
// on zeros, same binary).  A register-resident MFMA loop sustains 1.9 PF on random f16 data with 16x16x32 against 1.5-1.66 PF with
// 32x32x16 (tools/micro/mfma_power.hip, profiles/r04_mfma_power_shapes.txt): per FLOP the small shape moves 20 % fewer register-file
// bytes (A 4 + B 4 + C 4 + D 4 registers per 16 KFLOP against 4 + 4 + 16 + 16 per 32 KFLOP).  Same FLOPs per cycle on paper.
//
// What changes against ffn_fused.h:
//   * wave tile 64 ch x 64 frames = 4 x 4 fragments of 16 x 16, accumulators f32x4 [4][4] (same 64 + 64 registers); a phase (K = 32)
//     is 4 A-fragment reads + 4 B-fragment reads (ds_read_b128; as many LDS bytes as before) and 16 MFMAs.
//   * B fragments (h / u areas, [row][64 ch] 128-B rows, 16-B chunks XOR-swizzled by (row >> 1) & 7): lane l reads chunk
//     kp*4 + (l >> 4) of row base + j + ROWMAP(l & 15).  ds_read_b128 is served in four groups of 16 lanes that mix two k-groups
//     ({0-3, 12-15, 20-27}, ...): with rows in lane order every odd tap would be a 2-way bank conflict; with
//         ROWMAP(n) = 2n (n < 4), 2n - 7 (4 <= n < 12), 2n - 16 (n >= 12)
//     each group reads 8 even rows of one k-group and 8 odd rows of the other: conflict-free for all three taps (brute-forced over
//     every (wave, fragment, tap, k-step): tools/lds_bank_check.py).  Accumulator column n of frame fragment b therefore IS frame
//     b*16 + ROWMAP(n) -- the SiLU step, the mask and the epilogue's park use the same map.
//   * weight stream: slab = 16 A-fragments (channel quarter wq, fragment a) of 16 rows x K 32, lane-linear (common.h:
//     ffn_stream_index with stage bit 1): one address register, immediates for a, conflict-free by construction.
//   * 6 B-fragment address registers (tap x k-step pair) instead of 12.
// The accumulation order inside a fragment differs from the 32x32x16 path, so results are NOT bit-identical to the two-kernel
// path any more; they agree to fp32 accumulation noise (+ rare 1-ulp flips of the 16-bit u) -- tests/test_gpu_engine.py.
#pragma once
#include "ffn_fused.h"

namespace st {

__device__ __forceinline__ int ffn16_rowmap(int n) { return n < 4 ? 2 * n : (n < 12 ? 2 * n - 7 : 2 * n - 16); }

// ABL (developer ablations, tools only; results are garbage): 1 = no epilogue, 2 = no SiLU arithmetic, 32 = no MFMAs
template <class P, int ABL>
__global__ __launch_bounds__(512, 1)
void ffn_fused16_kernel(const ConvGemmArgs g) {
    constexpr int DEPTH = 3;
    using vec8 = typename P::vec8;
    constexpr int FV = kFfnFusedFrames, AREA = kFfnArea, SLAB = kFfnSlab, RING = kFfnRing;
    constexpr int OFF_RING = kFfnOffRing, OFF_BIAS = kFfnOffBias, OFF_SINK = kFfnOffSink;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, rm = ffn16_rowmap(lane & 15);
    const int grp = wave >> 2, ngrp = grp ^ 1;
    // group 0 = channel quarters 0, 1; group 1 = quarters 2, 3 (waves w and w + 4 share a SIMD)
    const int wc = ((wave >> 2) << 1) | ((wave >> 1) & 1), wf = wave & 1;
    const int T = g.T;
    const int nchunks = g.cmid >> 8;

    const int total = g.n_items * g.tiles_f;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int tf = lin % g.tiles_f, n = lin / g.tiles_f;
    const int t0 = tf * FV;
    if (g.t_lim && t0 >= g.t_lim[n % g.t_lim_mod]) return;      // ragged batch: this tile lies past the item's last needed frame

    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void_t*)smem);
    const unsigned char* h_s = sgpr_ptr64((const unsigned char*)g.a0 + (size_t)(n % g.a0_mod) * T * 512);
    const unsigned char* w_s = sgpr_ptr64(g.w);
    const unsigned char* zeros = (const unsigned char*)g.zeros;

    // ---- per-lane invariants -------------------------------------------------------------------------------------------
    // h pieces of this wave: pieces wave, wave + 8 and (wave 0 only) 16; 8 rows x 128 B each, row = frame t0 - 2 + row
    const int prow = lane >> 3;
    unsigned voffH[3]; bool validH[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pi = (k < 2) ? wave + 8 * k : 16;
        const int row = pi * 8 + prow;
        const int t = t0 - 2 + row;
        validH[k] = (t >= 0 && t < T) && (k < 2 || wave == 0);
        voffH[k] = (unsigned)(t * 512) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned voffL = (unsigned)lane * 16u;
    // B-fragment addresses (tap x k-step pair) inside area 0, frame fragment 0; area / fragment b are immediates (b * 16 rows)
    unsigned aadr[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const int row = wf * 64 + rm + j;
            aadr[j][kp] = lds0 + (unsigned)(row * 128 + (((kp * 4 + kg) ^ ((row >> 1) & 7)) << 4));
            asm volatile("" : "+v"(aadr[j][kp]));
        }
    unsigned wbase = lds0 + OFF_RING + voffL + (unsigned)wc * 4096u;      // A fragments (wc, a) of the slab at ring offset 0
    asm volatile("" : "+v"(wbase));
    // SiLU step: mask of this lane's four frame fragments, u addresses per channel fragment a (frame fragment b = immediate).
    // The lane's 4 accumulator rows are channels a*16 + 4*kg .. +4 of sub-chunk wc: 16-B chunk 2a + (kg >> 1), half kg & 1.
    float mk[4]; unsigned uadr[4];
    {
        const float* mrow = g.mask ? g.mask + (size_t)(n % g.mask_mod) * T : nullptr;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = wf * 64 + b * 16 + rm;
            const int t = t0 - 1 + row;
            const bool in = (t >= 0 && t < T);
            const float mv = mrow ? mrow[in ? t : 0] : 1.0f;
            mk[b] = in ? mv : 0.0f;                 // u outside [0, T) is conv_2's zero padding
        }
        const int row = wf * 64 + rm;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            uadr[a] = lds0 + (unsigned)(wc * AREA + row * 128 + (kg & 1) * 8 + (((2 * a + (kg >> 1)) ^ ((row >> 1) & 7)) << 4));
    }

    // ---- LDS-DMA issue ---------------------------------------------------------------------------------------------------
    auto issueH = [&](int ci, int k) {      // k: compile-time after unrolling
        const unsigned char* sb = h_s + ci * 128;
        const unsigned dst = (k < 2) ? lds0 + (unsigned)(ci * AREA + (wave + 8 * k) * 1024)
                                     : (wave == 0 ? lds0 + (unsigned)(ci * AREA + 16 * 1024) : lds0 + (unsigned)OFF_SINK);
        glds16bo(validH[k] ? sb + voffH[k] : zeros, dst);
    };
    unsigned roff = 0, woff = DEPTH * SLAB, sig = DEPTH;      // ring offsets of the slab being read / issued, index of the slab being issued
    auto issueW = [&]() {
        const unsigned char* sb = w_s + (size_t)sig * SLAB + (size_t)wave * 2048;
        const unsigned d = lds0 + (unsigned)OFF_RING + woff + (unsigned)wave * 2048u;
        glds16o(sb, voffL, d);
        glds16o(sb + 1024, voffL, d + 1024);
    };

    // ---- prologue: h chunks 0..2, slabs 0..2, conv_1 bias ------------------------------------------------------------------
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned dst = (k < 2) ? lds0 + (unsigned)(ci * AREA + (wave + 8 * k) * 1024)
                                         : (wave == 0 ? lds0 + (unsigned)(ci * AREA + 16 * 1024) : lds0 + (unsigned)OFF_SINK);
            glds16bo(validH[k] ? h_s + ci * 128 + voffH[k] : zeros, dst);
        }
#pragma unroll
    for (int sl = 0; sl < DEPTH; ++sl) {
        const unsigned char* sb = w_s + (size_t)sl * SLAB + (size_t)wave * 2048;
        const unsigned d = lds0 + (unsigned)(OFF_RING + sl * SLAB) + (unsigned)wave * 2048u;
        glds16o(sb, voffL, d);
        glds16o(sb + 1024, voffL, d + 1024);
    }
    if (wave < (g.cmid >> 8)) glds16o(sgpr_ptr64(g.bias1) + (size_t)wave * 1024, voffL, lds0 + (unsigned)OFF_BIAS + (unsigned)wave * 1024u);
    asm volatile("" :: "v"(mk[0]), "v"(mk[1]), "v"(mk[2]), "v"(mk[3]));      // the mask loads are waited for HERE, not inside the loop
    const bool allone = __all(mk[0] == 1.0f && mk[1] == 1.0f && mk[2] == 1.0f && mk[3] == 1.0f);
    ST_DMA_WAIT(0);
    __syncthreads();

    f32x4_t acc1[4][4], acc2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc2[a][b] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};

    vec8 wfr[4], bfr[4];
#define ST_BARRIER_IF(cond) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lnb_%=\n\ts_barrier\n.Lnb_%=:" :: "s"(cond) : "memory", "scc")
    // top-of-phase wait: everything but the pieces of the last DEPTH - 2 phases (ffn_allowed)
#define FF_TOPWAIT(LP) if (lastc) ffn_dma_wait<ffn_allowed((LP), true, DEPTH)>(); else ffn_dma_wait<ffn_allowed((LP), false, DEPTH)>();
#define FF_ISSUE(LP)                                                                             \
    if ((LP) + DEPTH < 48 || !lastc) issueW();                                                   \
    if constexpr ((LP) >= 1 && (LP) <= 3) issueH(3, (LP) - 1);                                   \
    if constexpr ((LP) >= 30 && ((LP) - 24) % 6 >= 1 && ((LP) - 24) % 6 <= 3) { if (!lastc) issueH(((LP) - 24) / 6 - 1, ((LP) - 24) % 6 - 1); }
#define FF_MMA(ACC)                                                                              \
    if constexpr (ABL & 32) { asm volatile("" :: "v"(wfr[0]), "v"(wfr[1]), "v"(wfr[2]), "v"(wfr[3]), "v"(bfr[0]), "v"(bfr[1]), "v"(bfr[2]), "v"(bfr[3])); } \
    else {                                                                                       \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                            \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) ACC[a][b] = P::mfma16(wfr[a], bfr[b], ACC[a][b]); \
    }
    // one phase = K 32: slab (ring offset roff), B fragments of area AR, tap J, k-step pair KP
#define FF_PHASE(ACC, AR, J, KP, LP)                                                             \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        FF_TOPWAIT(LP)                                                                           \
        ST_BARRIER_IF(grp);                                                                      \
        {                                                                                        \
            const unsigned wad = wbase + roff;                                                   \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) wfr[a] = as_vec8<P>(lds_read16(wad + a * 1024)); \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) bfr[b] = as_vec8<P>(lds_read16(aadr[J][KP] + (AR) * AREA + b * 2048)); \
        }                                                                                        \
        FF_ISSUE(LP)                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        ST_BARRIER_IF(ngrp);                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                           \
        FF_MMA(ACC)                                                                              \
        __builtin_amdgcn_s_setprio(0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        roff += SLAB; if (roff == RING * SLAB) roff = 0;                                         \
        woff += SLAB; if (woff == RING * SLAB) woff = 0;                                         \
        sig += 1;                                                                                \
    }
#define FF_STAGE6(ACC, AR, BASE)                                                                 \
    FF_PHASE(ACC, AR, 0, 0, (BASE) + 0) FF_PHASE(ACC, AR, 0, 1, (BASE) + 1)                      \
    FF_PHASE(ACC, AR, 1, 0, (BASE) + 2) FF_PHASE(ACC, AR, 1, 1, (BASE) + 3)                      \
    FF_PHASE(ACC, AR, 2, 0, (BASE) + 4) FF_PHASE(ACC, AR, 2, 1, (BASE) + 5)

#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const bool lastc = (c + 1 == nchunks);
        // acc1 = conv_1 bias of this chunk's channels
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 bv = *(const float4*)(smem + OFF_BIAS + (c * 256 + wc * 64 + a * 16 + 4 * kg) * 4);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc1[a][b] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
        }
        // ---- S1: conv_1, K = (cin chunk, tap, k-step pair); h chunk ci in area ci
        FF_STAGE6(acc1, 0, 0) FF_STAGE6(acc1, 1, 6) FF_STAGE6(acc1, 2, 12) FF_STAGE6(acc1, 3, 18)
        // every wave is done with h (the areas become u) ...
        __builtin_amdgcn_sched_barrier(0);
        ST_RAW_BARRIER();
        // ---- SiLU, mask, 16-bit rounding in the accumulator registers; u sub-chunk wc, rows of this wave, swizzled 128-B rows
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float m = mk[b];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4_t v = acc1[a][b];
                if constexpr (!(ABL & 2)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = silu_fast(v[r]);
                }
                if (!allone) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= m;
                }
                const uint2 pv = pack4<P>(v[0], v[1], v[2], v[3]);
                typedef unsigned __attribute__((ext_vector_type(2))) u32x2_raw;
                *(__attribute__((address_space(3))) u32x2_raw*)(uintptr_t)(uadr[a] + b * 2048) = u32x2_raw{pv.x, pv.y};
            }
        }
        // ... and u is complete before anyone reads it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ST_RAW_BARRIER();
        // ---- S2: conv_2, K = (u sub-chunk, tap, k-step pair); areas it has finished with are refilled with h for the next chunk
        FF_STAGE6(acc2, 0, 24) FF_STAGE6(acc2, 1, 30) FF_STAGE6(acc2, 2, 36) FF_STAGE6(acc2, 3, 42)
    }
#undef FF_STAGE6
#undef FF_PHASE
#undef FF_MMA
#undef FF_ISSUE
#undef FF_TOPWAIT
#undef ST_BARRIER_IF
    ST_DMA_WAIT(0);
    __syncthreads();
    if constexpr (ABL & 1) {
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" :: "v"(acc2[a][0]), "v"(acc2[a][1]), "v"(acc2[a][2]), "v"(acc2[a][3]));
        return;
    }
    // conv_2's EPI_RESGATE(+LayerNorm) epilogue; only the park of the accumulators knows about the fragment shape
    float* stage = (float*)smem;
    g2_epilogue_core<P, EPI_RESGATE, 256, 128, 4, 2>([&](int fbase) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int fl = fbase + b * 16 + rm;
                const int ch = wc * 64 + a * 16 + 4 * kg;
                *(float4*)(stage + fl * 260 + ch) = make_float4(acc2[a][b][0], acc2[a][b][1], acc2[a][b][2], acc2[a][b][3]);
            }
    }, stage, g, n, t0, FV, 0, wc + 4 * wf, lane);
}

template <class P>
static hipError_t launch_ffn_fused16_t(const ConvGemmArgs& a, hipStream_t s) {
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_done_dev[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)ffn_fused16_kernel<P, ST_FFN_ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds);
        if (e != hipSuccess) return e;
        attr_done_dev[dev_] = true;
    }
    if (!a.zeros || !a.w || !a.bias1 || a.cout != 256 || a.c0 != 256 || a.c1 || a.c2 || (a.cmid & 255) || a.cmid < 256 || a.cmid > 2048 ||
        a.ksplit > 1 || a.w_item_stride || a.branch32) return hipErrorInvalidValue;
    ConvGemmArgs b = a;
    b.tiles_f = (a.T + kFfnFusedFrames - 1) / kFfnFusedFrames;
    b.tiles_c = 1;
    const int total = b.n_items * b.tiles_f;
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL((ffn_fused16_kernel<P, ST_FFN_ABL>), dim3(grid), dim3(512), kFfnLds, s, b);
    return hipGetLastError();
}

}  // namespace st
