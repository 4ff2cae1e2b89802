// The FFN of a DiT block (diffusion_transformer.py:20-30: conv_1 k=3 256 -> F, SiLU, mask, conv_2 k=3 F -> 256) as ONE kernel whose
// F-wide intermediate `u` never leaves the CU.  As two kernels u is written (129 MB per layer at the headline size) and read back
// (131 MB): 15.6 of the 65.7 GB a solve moves, plus one HBM-bound epilogue and one block turnover per conv_1 tile.
//
// Block = 8 waves, one frame tile of 126 output frames of one item:
//   u rows r = 0..127   <-> frames t0-1+r   (the tile plus conv_2's halo)
//   h rows i = 0..129   <-> frames t0-2+i   (plus conv_1's halo); u row r, tap j reads h row r+j; output column f, tap j reads u row f+j
// The intermediate width is walked in CHUNKS of 256 channels.  Per chunk:
//   S1  acc1[256 ch x 128 rows] = bias_1 + conv_1 over K = 4 cin chunks x 3 taps x 64         (24 phases)
//       SiLU, mask, 16-bit rounding in the accumulator registers -> LDS, [row][64 ch] images per 64-channel sub-chunk
//   S2  acc2[256 ch x 128 cols] += conv_2 over K = 4 sub-chunks x 3 taps x 64                 (24 phases)
// and after the last chunk conv_2's ordinary EPI_RESGATE(+LayerNorm) epilogue (g2_epilogue).  Wave tile 64 ch x 64 frames in both
// stages (4 x 2 waves, 1.0 fragment reads per MFMA); acc1 + acc2 = 128 accumulator registers.  The K order of both contractions
// is the two-kernel path's (chunk, tap, k-step), SiLU / mask / rounding are the same instructions: results are BIT-IDENTICAL to
// conv_gemm_phased3_kernel<EPI_ACT16> followed by <EPI_RESGATE> (tests/test_gpu_engine.py).
//
// LDS (160,768 B):
//   4 operand AREAS of 136 rows x 128 B (17,408 B each).  During S1 area ci holds h's cin chunk ci (LDS-DMA, 17 pieces of 8 rows);
//     the SiLU step overwrites area s with u's sub-chunk s; while S2 walks the sub-chunks, the areas it has finished with are
//     refilled with h for the next chunk (area 3 during the next S1).  Same row pitch / XOR swizzle as the conv kernels, so ONE
//     set of 12 fragment addresses (tap x k-step) serves both stages; the area is an immediate offset.
//   weight RING of 5 slabs of 16 KiB.  The weights come as ONE linear stream in consumption order (launch_pack_ffn_stream): a
//     slab = one phase = K 32 x 256 rows = 16 MFMA A-fragments stored lane-linear, so a fragment read is base + lane*16 + immediate
//     (conflict-free, one address register) and the DMA source is contiguous.
//   conv_1 bias (F floats) and a 1-KiB sink for padding pieces.
// K loop = the phased scheme of conv_gemm_phased.h: a phase = 8 fragment reads + 8 MFMAs per wave + this phase's LDS-DMA issues,
// ONE raw s_barrier per phase, the two wave groups (waves 0-3 / 4-7, one of each per SIMD) half a phase apart: group 0 takes the
// barrier between its reads and its MFMAs, group 1 before its reads.  Slab p+3 is issued in phase p; every wave waits at the TOP
// of phase p with vmcnt(n_p), n_p = the pieces it issued in phase p-1, i.e. for everything up to slab p+1:
//   RAW  slab p+1 is first read by group 0 in phase p+1, after barrier #p, which every wave passes after its top-of-p wait.
//   WAR  slab p+3 overwrites slab p-2, last read by group 1 in the interval (p-2, p-1), retired (lgkmcnt 0) before its MFMAs, i.e.
//        before barrier #(p-1); group 0 issues after barrier #(p-1), group 1 after barrier #p.
// Every wave issues the same number of pieces in every phase (2 weight pieces; in 12 phases per chunk one h piece: 17 real pieces
// + 7 that read the zero page into the sink), so the counts hold for every wave.  Ordinary global loads (mask) happen before the
// loop only: they would count in vmcnt.
#pragma once
#include <type_traits>
#include "conv_gemm_phased.h"

namespace st {

constexpr int kFfnArea = 136 * 128, kFfnSlab = 16384, kFfnRing = 5;
constexpr int kFfnOffRing = 4 * kFfnArea, kFfnOffBias = kFfnOffRing + kFfnRing * kFfnSlab, kFfnOffSink = kFfnOffBias + 8192;
constexpr int kFfnLds = kFfnOffSink + 1024;      // 160,768 B

// LDS-DMA pieces a wave issues in local phase lp of a chunk (lp < 0: the previous, non-last chunk): 2 weight pieces unless the
// stream has ended, + 1 h piece in phases 1..3 (this chunk's h cin chunk 3 -> area 3) and, unless this is the last chunk, in
// phases 1..3 of S2's sub-chunks 1..3 (the next chunk's h cin chunks 0..2 -> the areas S2 has finished with)
constexpr bool ffn_w_issued(int lp, bool last, int depth) { return !last || lp + depth < 48; }
constexpr bool ffn_h_issued(int lp, bool last) { return (lp >= 1 && lp <= 3) || (!last && lp >= 30 && (lp - 24) % 6 >= 1 && (lp - 24) % 6 <= 3); }
constexpr int ffn_n_issued(int lp, bool last, int depth) {
    return lp < 0 ? ffn_n_issued(lp + 48, false, depth) : 2 * (int)ffn_w_issued(lp, last, depth) + (int)ffn_h_issued(lp, last);
}
// pieces that may stay outstanding at the top of phase lp: slab lp + 1 was issued in phase lp + 1 - depth, everything younger may fly
constexpr int ffn_allowed(int lp, bool last, int depth) {
    int n = 0;
    for (int k = 1; k <= depth - 2; ++k) n += ffn_n_issued(lp - k, last, depth);
    return n;
}
template <int N> __device__ __forceinline__ void ffn_dma_wait() {
    static_assert(N >= 0 && N <= 9, "vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}

// VAR: bit 0 = the phase's LDS-DMA issues sit between the two halves of its MFMAs (after the barrier) instead of in the read
//      segment; bit 1 = slab p + 4 instead of p + 3 is issued in phase p (legal only with bit 0: the slot it overwrites was last
//      read in the interval that ends with barrier #p).
// ABL (developer ablations, tools only; results are garbage): 1 = no epilogue, 2 = no SiLU arithmetic, 4 = the weight stream
//      re-reads its first slabs (cache-hot source), 8 = no top-of-phase waits, 16 = no LDS-DMA inside the loop, 32 = no MFMAs,
//      64 = s_memtime stamps per phase segment (g.dbg).  VAR bit 2 = s_setprio 1 around the MFMAs (the round-4 default; round 5: without it the solve is 0.4-0.8 % faster, paired -- profiles/r05_ab_setprio.txt), bit 3 = software-pipelined
//      phases without the group stagger (FF_PHASE_P).
template <class P, int ABL, int VAR>
__global__ __launch_bounds__(512, 1)
void ffn_fused_kernel(const ConvGemmArgs g) {
    constexpr int PLACE = VAR & 1, DEPTH = (VAR & 2) ? 4 : 3;
    static_assert(DEPTH == 3 || PLACE == 1 || (VAR & 8), "depth 4 needs the issue after the barrier");
    using vec8 = typename P::vec8;
    constexpr int FV = kFfnFusedFrames, AREA = kFfnArea, SLAB = kFfnSlab, RING = kFfnRing;
    constexpr int OFF_RING = kFfnOffRing, OFF_BIAS = kFfnOffBias, OFF_SINK = kFfnOffSink;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int grp = wave >> 2, ngrp = grp ^ 1;
    // group 0 = channel quarters 0, 1; group 1 = quarters 2, 3 (waves w and w + 4 share a SIMD)
    const int wc = ((wave >> 2) << 1) | ((wave >> 1) & 1), wf = wave & 1;
    const int T = g.T;
    const int nchunks = g.cmid >> 8;

    const int total = g.n_items * g.tiles_f;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // (tiles dealt round-robin to the XCDs instead: +0.3 % all-ones, -0.5 % ragged, paired -- profiles/r05_ab_prio_ffnrr.txt)
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
    // B-fragment addresses (tap x k-step) inside area 0, frame fragment 0; area / second fragment are immediates
    unsigned aadr[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = wf * 64 + l31 + j;
            aadr[j][ks] = lds0 + (unsigned)(row * 128 + (((ks * 2 + hi) ^ ((row >> 1) & 7)) << 4));
            asm volatile("" : "+v"(aadr[j][ks]));
        }
    unsigned wbase = lds0 + OFF_RING + voffL + (unsigned)wc * 2048u;      // A fragments (ksl, 2 wc + a) of the slab at ring offset 0
    asm volatile("" : "+v"(wbase));
    // SiLU step: u row / mask of this lane's two frame fragments
    float mk[2]; unsigned ubase[2], usw[2];
    {
        const float* mrow = g.mask ? g.mask + (size_t)(n % g.mask_mod) * T : nullptr;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int row = wf * 64 + b * 32 + l31;
            const int t = t0 - 1 + row;
            const bool in = (t >= 0 && t < T);
            const float mv = mrow ? mrow[in ? t : 0] : 1.0f;
            mk[b] = in ? mv : 0.0f;                 // u outside [0, T) is conv_2's zero padding
            ubase[b] = lds0 + (unsigned)(wc * AREA + row * 128 + hi * 8);
            usw[b] = (unsigned)(((row >> 1) & 7) << 4);
        }
    }

    // ---- LDS-DMA issue ---------------------------------------------------------------------------------------------------
    auto issueH = [&](int ci, int k) {      // k: compile-time after unrolling
        if constexpr (ABL & 16) return;
        const unsigned char* sb = h_s + ci * 128;
        const unsigned dst = (k < 2) ? lds0 + (unsigned)(ci * AREA + (wave + 8 * k) * 1024)
                                     : (wave == 0 ? lds0 + (unsigned)(ci * AREA + 16 * 1024) : lds0 + (unsigned)OFF_SINK);
        glds16bo(validH[k] ? sb + voffH[k] : zeros, dst);
    };
    unsigned roff = 0, woff = DEPTH * SLAB, sig = DEPTH;      // ring offsets of the slab being read / issued, index of the slab being issued
    auto issueW = [&]() {
        if constexpr (ABL & 16) return;
        const unsigned char* sb = w_s + (size_t)((ABL & 4) ? sig % 5 : sig) * SLAB + (size_t)wave * 2048;
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
    asm volatile("" :: "v"(mk[0]), "v"(mk[1]));      // the mask loads are waited for HERE, not at their first use inside the loop
    const bool allone0 = __all(mk[0] == 1.0f), allone1 = __all(mk[1] == 1.0f);
    ST_DMA_WAIT(0);
    __syncthreads();

    f32x16_t acc1[2][2], acc2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.0f;

    vec8 wfr[2][2], bfr[2][2];
    // ABL & 64: s_memtime stamps per phase segment -> g.dbg[block][wave 0 / 4][8] = top wait, barrier (group 1), reads + issue,
    // barrier (group 0), MFMA issue (to the next phase's top), SiLU step, -, total loop ticks
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, tstart = 0;
    if constexpr (ABL & 64) { tlast = tstart = __builtin_amdgcn_s_memtime(); }
#define ST_BARRIER_IF(cond) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lnb_%=\n\ts_barrier\n.Lnb_%=:" :: "s"(cond) : "memory", "scc")
    // top-of-phase wait: everything but the pieces of the last DEPTH - 2 phases (ffn_allowed)
#define FF_TOPWAIT(LP)                                                                           \
    if constexpr (!(ABL & 8)) {                                                                  \
        if (lastc) ffn_dma_wait<ffn_allowed((LP), true, DEPTH)>(); else ffn_dma_wait<ffn_allowed((LP), false, DEPTH)>(); \
    }
#define FF_ISSUE(LP)                                                                             \
    if ((LP) + DEPTH < 48 || !lastc) issueW();                                                   \
    if constexpr ((LP) >= 1 && (LP) <= 3) issueH(3, (LP) - 1);                                   \
    if constexpr ((LP) >= 30 && ((LP) - 24) % 6 >= 1 && ((LP) - 24) % 6 <= 3) { if (!lastc) issueH(((LP) - 24) / 6 - 1, ((LP) - 24) % 6 - 1); }
#define FF_MMA(ACC, KSL)                                                                         \
    if constexpr (ABL & 32) { asm volatile("" :: "v"(wfr[KSL][0]), "v"(wfr[KSL][1]), "v"(bfr[KSL][0]), "v"(bfr[KSL][1])); } \
    else {                                                                                       \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                            \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) ACC[a][b] = P::mfma(wfr[KSL][a], bfr[KSL][b], ACC[a][b]); \
    }
#define FF_STAMP(K) if constexpr (ABL & 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tm[K] += t_ - tlast; tlast = t_; }
#define FF_PHASE_S(ACC, AR, J, KP, LP)                                                           \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        FF_STAMP(4)                                                                              \
        FF_TOPWAIT(LP)                                                                           \
        FF_STAMP(0)                                                                              \
        ST_BARRIER_IF(grp);                                                                      \
        FF_STAMP(1)                                                                              \
        {                                                                                        \
            const unsigned wad = wbase + roff;                                                   \
            _Pragma("unroll") for (int ksl = 0; ksl < 2; ++ksl) {                                \
                _Pragma("unroll") for (int a = 0; a < 2; ++a) wfr[ksl][a] = as_vec8<P>(lds_read16(wad + ksl * 8192 + a * 1024)); \
                _Pragma("unroll") for (int b = 0; b < 2; ++b) bfr[ksl][b] = as_vec8<P>(lds_read16(aadr[J][2 * (KP) + ksl] + (AR) * AREA + b * 4096)); \
            }                                                                                    \
        }                                                                                        \
        if constexpr (!PLACE) { FF_ISSUE(LP) }                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        FF_STAMP(2)                                                                              \
        ST_BARRIER_IF(ngrp);                                                                     \
        FF_STAMP(3)                                                                              \
        if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(1);                                 \
        FF_MMA(ACC, 0)                                                                           \
        if constexpr (PLACE) {                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            FF_ISSUE(LP)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
        FF_MMA(ACC, 1)                                                                           \
        if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(0);                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        roff += SLAB; if (roff == RING * SLAB) roff = 0;                                         \
        woff += SLAB; if (woff == RING * SLAB) woff = 0;                                         \
        sig += 1;                                                                                \
    }
    // VAR & 8: software-pipelined phase, no group stagger.  Every wave: barrier, reads of the phase's second k-step, LDS-DMA issues,
    // MFMAs of the first k-step (its fragments were read during the previous phase), reads of the NEXT phase's first k-step, MFMAs of
    // the second k-step -- a wave's LDS latency hides under its own MFMAs.  The next slab is read one phase early: the top-of-phase
    // wait + barrier that cover slab p + 1 precede it; a slab's last read is finished before the barrier of the next phase, so the
    // ring's WAR distance is one barrier (DEPTH <= RING - 1).  Stage starts (phases 0 and 24) load both halves.
#define FF_LOAD_HALF(KSL, ROFF, AR, J, KP)                                                       \
    {                                                                                            \
        const unsigned wad_ = wbase + (ROFF);                                                    \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) wfr[KSL][a] = as_vec8<P>(lds_read16(wad_ + (KSL) * 8192 + a * 1024)); \
        _Pragma("unroll") for (int b = 0; b < 2; ++b) bfr[KSL][b] = as_vec8<P>(lds_read16(aadr[J][2 * (KP) + (KSL)] + (AR) * AREA + b * 4096)); \
    }
#define FF_PHASE_P(ACC, AR, J, KP, LP)                                                           \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        FF_TOPWAIT(LP)                                                                           \
        ST_RAW_BARRIER();                                                                        \
        if constexpr ((LP) == 0 || (LP) == 24) FF_LOAD_HALF(0, roff, AR, J, KP)                  \
        FF_LOAD_HALF(1, roff, AR, J, KP)                                                         \
        FF_ISSUE(LP)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(1);                                 \
        FF_MMA(ACC, 0)                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        roff += SLAB; if (roff == RING * SLAB) roff = 0;                                         \
        woff += SLAB; if (woff == RING * SLAB) woff = 0;                                         \
        sig += 1;                                                                                \
        if constexpr ((LP) != 23 && (LP) != 47) FF_LOAD_HALF(0, roff, (((LP) + 1) % 24) / 6, (((LP) + 1) % 6) / 2, ((LP) + 1) & 1) \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        FF_MMA(ACC, 1)                                                                           \
        if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(0);                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
#define FF_PHASE(ACC, AR, J, KP, LP)                                                             \
    if constexpr (VAR & 8) FF_PHASE_P(ACC, AR, J, KP, LP) else FF_PHASE_S(ACC, AR, J, KP, LP)
#define FF_STAGE6(ACC, AR, BASE)                                                                 \
    FF_PHASE(ACC, AR, 0, 0, (BASE) + 0) FF_PHASE(ACC, AR, 0, 1, (BASE) + 1)                      \
    FF_PHASE(ACC, AR, 1, 0, (BASE) + 2) FF_PHASE(ACC, AR, 1, 1, (BASE) + 3)                      \
    FF_PHASE(ACC, AR, 2, 0, (BASE) + 4) FF_PHASE(ACC, AR, 2, 1, (BASE) + 5)

#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const bool lastc = (c + 1 == nchunks);
        // acc1 = conv_1 bias of this chunk's channels (the two-kernel path feeds it as the C operand of the first MFMAs: same sum)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 bv = *(const float4*)(smem + OFF_BIAS + (c * 256 + wc * 64 + a * 32 + 8 * q4 + 4 * hi) * 4);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc1[a][b][4 * q4 + 0] = bv.x; acc1[a][b][4 * q4 + 1] = bv.y;
                    acc1[a][b][4 * q4 + 2] = bv.z; acc1[a][b][4 * q4 + 3] = bv.w;
                }
            }
        // ---- S1: conv_1, K = (cin chunk, tap, k-step); h chunk ci in area ci
        FF_STAGE6(acc1, 0, 0) FF_STAGE6(acc1, 1, 6) FF_STAGE6(acc1, 2, 12) FF_STAGE6(acc1, 3, 18)
        // every wave is done with h (the areas become u) ...
        __builtin_amdgcn_sched_barrier(0);
        ST_RAW_BARRIER();
        // ---- SiLU, mask, 16-bit rounding in the accumulator registers; u sub-chunk wc, rows of this wave, swizzled 128-B rows
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float m = mk[b];
            const bool allone = b ? allone1 : allone0;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x16_t v = acc1[a][b];
                if constexpr (!(ABL & 2)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = silu_fast(v[r]);
                }
                if (!allone) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] *= m;
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const unsigned addr = ubase[b] + ((unsigned)((a * 4 + q4) << 4) ^ usw[b]);
                    const uint2 pv = pack4<P>(v[4 * q4 + 0], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
                    typedef unsigned __attribute__((ext_vector_type(2))) u32x2_raw;
                    *(__attribute__((address_space(3))) u32x2_raw*)(uintptr_t)addr = u32x2_raw{pv.x, pv.y};
                }
            }
        }
        // ... and u is complete before anyone reads it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ST_RAW_BARRIER();
        FF_STAMP(5)
        // ---- S2: conv_2, K = (u sub-chunk, tap, k-step); areas it has finished with are refilled with h for the next chunk
        FF_STAGE6(acc2, 0, 24) FF_STAGE6(acc2, 1, 30) FF_STAGE6(acc2, 2, 36) FF_STAGE6(acc2, 3, 42)
    }
    if constexpr (ABL & 64) {
        if (g.dbg && lane == 0 && (wave & 3) == 0 && lin < 64) {
            unsigned long long* d = g.dbg + (size_t)(lin * 2 + (wave >> 2)) * 8;
            for (int k = 0; k < 6; ++k) d[k] = tm[k];
            d[6] = 0; d[7] = __builtin_amdgcn_s_memtime() - tstart;
        }
    }
#undef FF_STAMP
#undef FF_STAGE6
#undef FF_PHASE
#undef FF_PHASE_S
#undef FF_PHASE_P
#undef FF_LOAD_HALF
#undef FF_MMA
#undef FF_ISSUE
#undef FF_TOPWAIT
#undef ST_BARRIER_IF
    ST_DMA_WAIT(0);
    __syncthreads();
    if constexpr (ABL & 1) {
        asm volatile("" :: "v"(acc2[0][0]), "v"(acc2[0][1]), "v"(acc2[1][0]), "v"(acc2[1][1]));
        return;
    }
    g2_epilogue<P, EPI_RESGATE, 256, 128, 4, 2>(acc2, (float*)smem, g, n, t0, FV, 0, wc + 4 * wf, lane);
}

#if defined(ST_FFN_ABL) && !defined(ST_DEVTOOLS)
#error "ST_FFN_ABL (ablation builds: results are garbage) needs -DST_DEVTOOLS"
#endif
#ifndef ST_FFN_ABL
#define ST_FFN_ABL 0
#endif
#ifndef ST_FFN_VAR
#define ST_FFN_VAR 0
#endif

template <class P>
static hipError_t launch_ffn_fused_t(const ConvGemmArgs& a, hipStream_t s) {
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_done_dev[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)ffn_fused_kernel<P, ST_FFN_ABL, ST_FFN_VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds);
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
    hipLaunchKernelGGL((ffn_fused_kernel<P, ST_FFN_ABL, ST_FFN_VAR>), dim3(grid), dim3(512), kFfnLds, s, b);
    return hipGetLastError();
}

}  // namespace st
