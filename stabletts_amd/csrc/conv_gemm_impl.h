// Implicit-GEMM Conv1d (k = 1 or 3, "same" zero padding) on MFMA 32x32x16 for gfx950.
//
// Replaces the nn.Conv1d calls of the reference hot path (models/estimator.py:78-94,
// models/diffusion_transformer.py:20-21,43-51).  Orientation: the MFMA A operand is the weight
// tile (rows = output channels), the B operand is the activation tile (cols = frames), so the
// accumulator is C[channel][frame] with one FRAME PER LANE: per-frame quantities (mask, RoPE
// angle, LayerNorm statistics) are lane-local, per-channel ones are 4-wide vectors.
//
// Block tile: BC=128 output channels x BF=128 frames, 4 waves (2 x 2), each wave 64 x 64 =
// 2 x 2 MFMA fragments (64 accumulator VGPRs).  K loop: 64 input channels per stage; for k=3
// the activation tile (BF + 2 halo frames) is staged ONCE per channel chunk and the three taps
// read it at row offsets 0/1/2, so activations cross HBM->LDS once, not three times.
// Staging: global -> registers (issued before the MFMAs) -> LDS (after them), double-buffered,
// one barrier per K stage.  LDS rows are padded to 144 B: 16 lanes of a ds_read_b128 group hit
// 16 distinct rows mod 16 -> conflict-free.  2 blocks/CU (74 KB LDS each).
#pragma once
#include "common.h"
#include "launch.h"
#include <cstdlib>
#include <type_traits>

namespace st {

constexpr int kBC = 128, kBF = 128, kWC = 2, kWF = 2;

// Accumulator tile -> global memory.  C[channel][frame]: lane = frame, 4-wide channel vectors.
template <class P, int EPI>
__device__ __forceinline__ void conv_epilogue(f32x16_t (&acc)[kBC / kWC / 32][kBF / kWF / 32], const ConvGemmArgs& g,
                                              int n, int t0, int cbase, int wc, int wf, int l31, int hi) {
    constexpr int BC = kBC, BF = kBF, WC = kWC, WF = kWF;
    constexpr int FC = BC / WC / 32, FF = BF / WF / 32;
    const int T = g.T;
    // ------------------------------------------------------------------ epilogue
    const float* mrow = g.mask ? g.mask + (size_t)(n % g.mask_mod) * T : nullptr;
#pragma unroll
    for (int b = 0; b < FF; ++b) {
        const int t = t0 + wf * (BF / WF) + b * 32 + l31;
        const bool tv = t < T;
        const float m = (mrow && tv) ? mrow[t] : 1.0f;
        const size_t grow = (size_t)n * T + t;
#pragma unroll
        for (int a = 0; a < FC; ++a) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int chl = wc * (BC / WC) + a * 32 + 8 * q4 + 4 * hi;   // channel within block tile
                const int ch = cbase + chl;
                float v0 = acc[a][b][4 * q4 + 0], v1 = acc[a][b][4 * q4 + 1];
                float v2 = acc[a][b][4 * q4 + 2], v3 = acc[a][b][4 * q4 + 3];
                if (g.bias) {
                    const float4 bb = *(const float4*)(g.bias + ch);
                    v0 += bb.x; v1 += bb.y; v2 += bb.z; v3 += bb.w;
                }
                if constexpr (EPI == EPI_ACT16) {
                    if (g.flags & GF_SILU) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                    if (g.flags & GF_MASK) { v0 *= m; v1 *= m; v2 *= m; v3 *= m; }
                    if (tv) *(uint2*)((unsigned char*)g.out16 + (grow * g.cout + ch) * 2) = pack4<P>(v0, v1, v2, v3);
                } else if constexpr (EPI == EPI_F32) {
                    if (tv) {
                        if (g.add32) {
                            const int an = n < g.add_clamp ? n : g.add_clamp;
                            const float4 ad = *(const float4*)(g.add32 + ((size_t)an * T + t) * g.cout + ch);
                            v0 += ad.x; v1 += ad.y; v2 += ad.z; v3 += ad.w;
                        }
                        if (g.flags & GF_MASK) { v0 *= m; v1 *= m; v2 *= m; v3 *= m; }
                        if (g.out32) *(float4*)(g.out32 + grow * g.cout + ch) = make_float4(v0, v1, v2, v3);
                        if (g.out16) *(uint2*)((unsigned char*)g.out16 + (grow * g.cout + ch) * 2) = pack4<P>(v0, v1, v2, v3);
                    }
                } else if constexpr (EPI == EPI_RESGATE) {
                    if (tv) {
                        const float4 gt = *(const float4*)(g.gate + (size_t)n * g.gate_stride + ch);
                        float4 x = *(const float4*)(g.out32 + grow * g.cout + ch);
                        x.x += gt.x * (v0 * m); x.y += gt.y * (v1 * m);
                        x.z += gt.z * (v2 * m); x.w += gt.w * (v3 * m);
                        *(float4*)(g.out32 + grow * g.cout + ch) = x;
                        if (g.out16) *(uint2*)((unsigned char*)g.out16 + (grow * g.cout + ch) * 2) = pack4<P>(x.x, x.y, x.z, x.w);
                    }
                } else {  // EPI_QKV: stash in registers, RoPE below
                    acc[a][b][4 * q4 + 0] = v0; acc[a][b][4 * q4 + 1] = v1;
                    acc[a][b][4 * q4 + 2] = v2; acc[a][b][4 * q4 + 3] = v3;
                }
            }
        }
        if constexpr (EPI == EPI_QKV) {
            // This wave's 64 channels are exactly one head of q, k or v (BC/WC == 64, C == 256):
            // fragment a=0 holds head dims 0..31 (the rotary half), a=1 holds 32..63 (pass-through).
            const int C = g.cout / 3;
            const int chw = cbase + wc * 64;          // first channel of this wave
            const int which = chw / C;                // 0 q, 1 k, 2 v
            const int head = (chw % C) >> 6;
            const int H = g.n_heads;
            if (which < 2) {
                // partial RoPE (diffusion_transformer.py:180-198): pairs (j, j+16), j < 16, in the
                // accumulator these are register groups q4 and q4+2 of fragment 0 -- lane local.
                if (tv) {
#pragma unroll
                    for (int q4 = 0; q4 < 2; ++q4) {
                        const int j0 = 8 * q4 + 4 * hi;
                        const float4 cs = *(const float4*)(g.rope_cos + (size_t)t * 16 + j0);
                        const float4 sn = *(const float4*)(g.rope_sin + (size_t)t * 16 + j0);
                        const float cc[4] = {cs.x, cs.y, cs.z, cs.w};
                        const float ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x1 = acc[0][b][4 * q4 + e];
                            const float x2 = acc[0][b][4 * (q4 + 2) + e];
                            acc[0][b][4 * q4 + e] = x1 * cc[e] - x2 * ss[e];
                            acc[0][b][4 * (q4 + 2) + e] = x2 * cc[e] + x1 * ss[e];
                        }
                    }
                    const float sc = (which == 0) ? g.qscale : 1.0f;
                    unsigned char* dst = (unsigned char*)(which == 0 ? g.q : g.k) +
                                         (((size_t)n * H + head) * T + t) * 64 * 2;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const int d = a * 32 + 8 * q4 + 4 * hi;
                            *(uint2*)(dst + d * 2) = pack4<P>(acc[a][b][4 * q4 + 0] * sc, acc[a][b][4 * q4 + 1] * sc,
                                                              acc[a][b][4 * q4 + 2] * sc, acc[a][b][4 * q4 + 3] * sc);
                        }
                }
            } else {
                // V is stored transposed [head][d][Tp] (keys contiguous) for the PV MFMA operand;
                // frames in [T, Tp) are written as zeros so attention can load whole 64-key tiles.
                if (t < g.Tp) {
                    // key order inside each group of 16: bits 2 and 3 swapped (PV operand order, attention.hip)
                    const int tpos = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
                    typename P::elem* dst = (typename P::elem*)g.vt + ((size_t)n * H + head) * 64 * g.Tp + tpos;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int d = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            dst[(size_t)d * g.Tp] = to16<P>(tv ? acc[a][b][r] : 0.0f);
                        }
                }
            }
        }
    }
}

// VAR 0: prefetch distance 1 (one register set, global->regs issued before the MFMAs of the same stage).
// VAR 1: prefetch distance 2 (two register sets for the per-stage tiles; the k=3 activation tile, which is
//        only needed every third stage, is issued at tap 0 and written to LDS at tap 2): every global load
//        has a full stage of MFMAs between issue and first use, so HBM/L2 latency leaves the critical path.
template <class P, int TAPS, int EPI, int VAR>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvGemmArgs g) {
    using vec8 = typename P::vec8;
    constexpr int BC = kBC, BF = kBF, WC = kWC, WF = kWF;
    constexpr int NT = 64 * WC * WF;
    constexpr int AROWS = BF + TAPS - 1;
    constexpr int ROWB = kLdsRowBytes;
    constexpr int A_BYTES = AROWS * ROWB;
    constexpr int W_BYTES = BC * ROWB;
    constexpr int NA = (AROWS * 8 + NT - 1) / NT;
    constexpr int NW = (BC * 8) / NT;
    constexpr int FC = BC / WC / 32, FF = BF / WF / 32;
    static_assert((BC * 8) % NT == 0, "weight tile must split evenly");
    static_assert(EPI != EPI_QKV || (BC / WC == 64), "QKV epilogue needs one head per wave");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Ws = smem + 2 * A_BYTES;

    // XCD-aware tile order: blocks with equal (blockIdx & 7) share an XCD/L2; give each XCD a
    // contiguous run of tiles with the channel tile fastest so an activation tile is re-read
    // from L2 by its tiles_c consumers.
    const int total = g.n_items * g.tiles_f * g.tiles_c;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int tc = lin % g.tiles_c;
    const int rest = lin / g.tiles_c;
    const int tf = rest % g.tiles_f;
    const int n = rest / g.tiles_f;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wave % WC, wf = wave / WC;
    const int cbase = tc * BC, t0 = tf * BF;
    const int cin = g.c0 + g.c1;
    const int nch = cin >> 6;
    const int T = g.T;

    const unsigned char* a0 = (const unsigned char*)g.a0 + (size_t)(n % g.a0_mod) * T * g.c0 * 2;
    const unsigned char* a1 = g.c1 ? (const unsigned char*)g.a1 + (size_t)(n % g.a1_mod) * T * g.c1 * 2 : nullptr;
    const unsigned char* wsrc = (const unsigned char*)g.w;

    auto loadA = [&](uint4 (&ra)[NA], int c) {
        const int ch0 = c << 6;
        const unsigned char* src; int cs, coff;
        if (ch0 < g.c0) { src = a0; cs = g.c0; coff = ch0; }
        else            { src = a1; cs = g.c1; coff = ch0 - g.c0; }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * NT;
            const int row = idx >> 3, seg = idx & 7;
            const int t = t0 + row - (TAPS / 2);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < AROWS * 8 && t >= 0 && t < T) {
                if constexpr (VAR >= 5)   // EXPERIMENT: chunk-major activations [item][chunk][T][64]
                    v = *(const uint4*)(src + ((size_t)(coff >> 6) * T + t) * 128 + seg * 16);
                else
                    v = *(const uint4*)(src + ((size_t)t * cs + coff) * 2 + seg * 16);
            }
            ra[i] = v;
        }
    };
    auto storeA = [&](const uint4 (&ra)[NA], int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * NT;
            const int row = idx >> 3, seg = idx & 7;
            if (idx < AROWS * 8) *(uint4*)(As + buf * A_BYTES + row * ROWB + seg * 16) = ra[i];
        }
    };
    auto loadW = [&](uint4 (&rw)[NW], int c, int j) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + i * NT;
            const int row = idx >> 3, seg = idx & 7;
            if constexpr (VAR >= 5)       // EXPERIMENT: tile-major weights [tc][chunk][tap][128][64]
                rw[i] = *(const uint4*)(wsrc + ((size_t)((tc * nch + c) * TAPS + j) * BC * 128) + idx * 16);
            else
                rw[i] = *(const uint4*)(wsrc + ((size_t)((cbase + row) * TAPS + j) * cin + (c << 6)) * 2 + seg * 16);
        }
    };
    auto storeW = [&](const uint4 (&rw)[NW], int buf) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + i * NT;
            const int row = idx >> 3, seg = idx & 7;
            *(uint4*)(Ws + buf * W_BYTES + row * ROWB + seg * 16) = rw[i];
        }
    };

    f32x16_t acc[FC][FF];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FF; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto compute = [&](int abuf, int wbuf, int j) {
        const unsigned char* Ab = As + abuf * A_BYTES + (wf * (BF / WF) + l31 + j) * ROWB + hi * 16;
        const unsigned char* Wb = Ws + wbuf * W_BYTES + (wc * (BC / WC) + l31) * ROWB + hi * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            vec8 wfr[FC], afr[FF];
#pragma unroll
            for (int a = 0; a < FC; ++a) wfr[a] = as_vec8<P>(*(const uint4*)(Wb + a * 32 * ROWB + ks * 32));
#pragma unroll
            for (int b = 0; b < FF; ++b) afr[b] = as_vec8<P>(*(const uint4*)(Ab + b * 32 * ROWB + ks * 32));
#pragma unroll
            for (int a = 0; a < FC; ++a)
#pragma unroll
                for (int b = 0; b < FF; ++b) acc[a][b] = P::mfma(wfr[a], afr[b], acc[a][b]);
        }
    };

    if constexpr (VAR >= 7 && VAR <= 9) {
        // ABLATIONS, memory side only (no MFMA): 7 = W tiles only, 8 = A tiles only, 9 = both but no LDS stores
        uint4 ra[NA], rw[NW];
        loadA(ra, 0); loadW(rw, 0, 0);
        storeA(ra, 0); storeW(rw, 0);
        __syncthreads();
        int it = 0;
        for (int c = 0; c < nch; ++c) {
#pragma unroll
            for (int j = 0; j < TAPS; ++j) {
                const bool last = (c == nch - 1) && (j == TAPS - 1);
                const bool nextA = (j == 0) && (c + 1 < nch);
                if (VAR != 7 && nextA) loadA(ra, c + 1);
                if (VAR != 8 && !last) { if (j == TAPS - 1) loadW(rw, c + 1, 0); else loadW(rw, c, j + 1); }
                if (VAR == 9) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) asm volatile("" :: "v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w));
#pragma unroll
                    for (int i = 0; i < NW; ++i) asm volatile("" :: "v"(rw[i].x), "v"(rw[i].y), "v"(rw[i].z), "v"(rw[i].w));
                } else {
                    if (VAR != 7 && nextA) storeA(ra, (c + 1) & 1);
                    if (VAR != 8 && !last) storeW(rw, (it + 1) & 1);
                }
                __syncthreads();
                ++it;
            }
        }
    } else if constexpr (VAR == 2 || VAR == 3 || VAR == 4) {
        // ABLATIONS (tools/gemm_bench only; results are wrong by construction):
        //   2: no global loads / LDS stores in the loop (LDS-read + MFMA + barrier only)
        //   3: loads + LDS stores + barrier, no MFMA      4: like 2 but also no barrier
        uint4 ra[NA], rw[NW];
        loadA(ra, 0); loadW(rw, 0, 0);
        storeA(ra, 0); storeW(rw, 0); storeA(ra, 1); storeW(rw, 1);
        __syncthreads();
        int it = 0;
        for (int c = 0; c < nch; ++c) {
#pragma unroll
            for (int j = 0; j < TAPS; ++j) {
                const bool last = (c == nch - 1) && (j == TAPS - 1);
                const bool nextA = (j == 0) && (c + 1 < nch);
                if (VAR == 3) {
                    if (nextA) loadA(ra, c + 1);
                    if (!last) { if (j == TAPS - 1) loadW(rw, c + 1, 0); else loadW(rw, c, j + 1); }
                    if (nextA) storeA(ra, (c + 1) & 1);
                    if (!last) storeW(rw, (it + 1) & 1);
                } else {
                    compute(c & 1, it & 1, j);
                }
                if (VAR != 4) __syncthreads();
                ++it;
            }
        }
    } else if constexpr (VAR == 0 || VAR == 5) {
        uint4 ra[NA], rw[NW];
        loadA(ra, 0); loadW(rw, 0, 0);
        storeA(ra, 0); storeW(rw, 0);
        __syncthreads();
        int it = 0;
        for (int c = 0; c < nch; ++c) {
#pragma unroll
            for (int j = 0; j < TAPS; ++j) {
                const bool last = (c == nch - 1) && (j == TAPS - 1);
                const bool nextA = (j == 0) && (c + 1 < nch);
                if (nextA) loadA(ra, c + 1);
                if (!last) { if (j == TAPS - 1) loadW(rw, c + 1, 0); else loadW(rw, c, j + 1); }
                compute(c & 1, it & 1, j);
                if (nextA) storeA(ra, (c + 1) & 1);
                if (!last) storeW(rw, (it + 1) & 1);
                __syncthreads();
                ++it;
            }
        }
    } else if constexpr (TAPS == 3) {
        // stage g = 3c + j holds W(c, j); invariant at the top of stage g: Wbuf[g&1] = W(g) in LDS,
        // the "old" register set carries W(g+1) (in flight), the "new" set is free for W(g+2).
        uint4 ra[NA], rwA[NW], rwB[NW];
        const int nst = nch * 3;
        loadA(ra, 0); loadW(rwA, 0, 0);
        storeA(ra, 0); storeW(rwA, 0);
        loadW(rwB, 0, 1);
        __syncthreads();
        auto stage = [&](auto jc, int c, uint4 (&rw_new)[NW], const uint4 (&rw_old)[NW], int wbuf, int abuf) {
            constexpr int J = decltype(jc)::value;
            const int gidx = c * 3 + J;
            if (gidx + 2 < nst) { if (J == 0) loadW(rw_new, c, 2); else loadW(rw_new, c + 1, J - 1); }
            if (J == 0 && c + 1 < nch) loadA(ra, c + 1);
            compute(abuf, wbuf, J);
            if (gidx + 1 < nst) storeW(rw_old, wbuf ^ 1);
            if (J == 2 && c + 1 < nch) storeA(ra, abuf ^ 1);
            __syncthreads();
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int c = 0; c < nch; c += 2) {          // nch is even (host guarantees cin % 128 == 0)
            stage(I0{}, c, rwA, rwB, 0, 0);
            stage(I1{}, c, rwB, rwA, 1, 0);
            stage(I2{}, c, rwA, rwB, 0, 0);
            stage(I0{}, c + 1, rwB, rwA, 1, 1);
            stage(I1{}, c + 1, rwA, rwB, 0, 1);
            stage(I2{}, c + 1, rwB, rwA, 1, 1);
        }
    } else {
        // TAPS == 1: stage g = c needs A(c) and W(c); both run two stages ahead in registers.
        uint4 raA[NA], raB[NA], rwA[NW], rwB[NW];
        loadA(raA, 0); loadW(rwA, 0, 0);
        storeA(raA, 0); storeW(rwA, 0);
        if (nch > 1) { loadA(raB, 1); loadW(rwB, 1, 0); }
        __syncthreads();
        auto stage = [&](int c, uint4 (&ra_new)[NA], const uint4 (&ra_old)[NA], uint4 (&rw_new)[NW],
                         const uint4 (&rw_old)[NW], int buf) {
            if (c + 2 < nch) { loadA(ra_new, c + 2); loadW(rw_new, c + 2, 0); }
            compute(buf, buf, 0);
            if (c + 1 < nch) { storeA(ra_old, buf ^ 1); storeW(rw_old, buf ^ 1); }
            __syncthreads();
        };
        for (int c = 0; c < nch; c += 2) {
            stage(c, raA, raB, rwA, rwB, 0);
            stage(c + 1, raB, raA, rwB, rwA, 1);
        }
    }

    conv_epilogue<P, EPI>(acc, g, n, t0, cbase, wc, wf, l31, hi);
}

// ------------------------------------------------------------------------------------------
// Direct-to-LDS variant: every tile byte goes HBM/L2 -> LDS through global_load_lds_dwordx4 (no
// staging VGPRs, no ds_write instructions).  An LDS-DMA instruction writes wave-uniform base +
// lane*16, so the LDS image is dense (128-B rows, no padding); bank conflicts are removed by an
// XOR swizzle applied on the SOURCE address and again on the fragment reads (guide rule 21):
//      LDS slot (row, s')  holds tile segment  s = s' ^ ((row >> 1) & 7)
// 16 rows that are distinct mod 16 then cover all sixteen 16-byte bank slots -> conflict-free
// ds_read_b128.  Out-of-range halo frames read a 16-byte zero block in global memory.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) { glds16b(gsrc, lds_wave_base); }

// OPT (experiments, tools/gemm_bench): 1 = s_setprio around the MFMA cluster, 2 = no LDS-DMA in the loop,
// 4 = no MFMA, 8 = single activation buffer + 3 blocks/CU (racy: timing only).
template <class P, int TAPS, int EPI, int OPT = 0>
__global__ __launch_bounds__(256, (OPT & 8) ? 3 : 2) void conv_gemm_glds_kernel(const ConvGemmArgs g) {
    using vec8 = typename P::vec8;
    constexpr int BC = kBC, BF = kBF, WC = kWC, WF = kWF;
    constexpr int AROWS = BF + TAPS - 1;
    constexpr int A_BYTES = AROWS * 128;
    constexpr int W_BYTES = BC * 128;
    constexpr int FC = BC / WC / 32, FF = BF / WF / 32;
    static_assert(BC == 128 && BF == 128 && WC * WF == 4, "tile constants");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ABUFS = (OPT & 8) ? 1 : 2;
    unsigned char* As = smem;
    unsigned char* Ws = smem + ABUFS * A_BYTES;

    const int total = g.n_items * g.tiles_f * g.tiles_c;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int tc = lin % g.tiles_c;
    const int rest = lin / g.tiles_c;
    const int tf = rest % g.tiles_f;
    const int n = rest / g.tiles_f;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wave % WC, wf = wave / WC;
    const int cbase = tc * BC, t0 = tf * BF;
    const int cin = g.c0 + g.c1;
    const int nch = cin >> 6;
    const int T = g.T;

    const unsigned char* a0 = (const unsigned char*)g.a0 + (size_t)(n % g.a0_mod) * T * g.c0 * 2;
    const unsigned char* a1 = g.c1 ? (const unsigned char*)g.a1 + (size_t)(n % g.a1_mod) * T * g.c1 * 2 : nullptr;
    const unsigned char* wsrc = (const unsigned char*)g.w;
    const unsigned char* zeros = (const unsigned char*)g.zeros;

    // this lane's (row, source segment) inside each 1-KiB piece: piece p covers rows 8p .. 8p+7
    const int prow = lane >> 3;                         // row within the piece
    // wave w issues pieces w*4 .. w*4+3 of a 128-row tile
    auto issueW = [&](int c, int j, int buf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = wave * 4 + k;
            const int row = piece * 8 + prow;
            const int seg = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned char* src = wsrc + ((size_t)((cbase + row) * TAPS + j) * cin + (c << 6)) * 2 + seg * 16;
            glds16(src, Ws + buf * W_BYTES + piece * 1024);
        }
    };
    auto issueA = [&](int c, int buf) {
        const int ch0 = c << 6;
        const unsigned char* srcb; int cs, coff;
        if (ch0 < g.c0) { srcb = a0; cs = g.c0; coff = ch0; }
        else            { srcb = a1; cs = g.c1; coff = ch0 - g.c0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = wave * 4 + k;
            const int row = piece * 8 + prow;
            const int seg = (lane & 7) ^ ((row >> 1) & 7);
            const int t = t0 + row - (TAPS / 2);
            const unsigned char* src = (t >= 0 && t < T) ? srcb + ((size_t)t * cs + coff) * 2 + seg * 16 : zeros;
            glds16(src, As + (buf % ABUFS) * A_BYTES + piece * 1024);
        }
        if constexpr (TAPS == 3) {
            // halo rows 128, 129: one partial piece (16 lanes) issued by wave 0
            if (wave == 0 && lane < 16) {
                const int row = 128 + prow;
                const int seg = (lane & 7) ^ ((row >> 1) & 7);
                const int t = t0 + row - 1;
                const unsigned char* src = (t < T) ? srcb + ((size_t)t * cs + coff) * 2 + seg * 16 : zeros;
                glds16(src, As + (buf % ABUFS) * A_BYTES + 16 * 1024);
            }
        }
    };

    f32x16_t acc[FC][FF];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FF; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment read addressing: byte = row*128 + ((ks*2 + hi) ^ ((row>>1)&7))*16
    int wrow_off[FC], wswz[FC];
#pragma unroll
    for (int a = 0; a < FC; ++a) {
        const int row = wc * (BC / WC) + a * 32 + l31;
        wrow_off[a] = row * 128; wswz[a] = (row >> 1) & 7;
    }
    auto compute = [&](int abuf, int wbuf, int j) {
        const unsigned char* Ab = As + (abuf % ABUFS) * A_BYTES;
        const unsigned char* Wb = Ws + wbuf * W_BYTES;
        int arow_off[FF], aswz[FF];
#pragma unroll
        for (int b = 0; b < FF; ++b) {
            const int row = wf * (BF / WF) + b * 32 + l31 + j;
            arow_off[b] = row * 128; aswz[b] = (row >> 1) & 7;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            vec8 wfr[FC], afr[FF];
            const int seg = ks * 2 + hi;
#pragma unroll
            for (int a = 0; a < FC; ++a) wfr[a] = as_vec8<P>(*(const uint4*)(Wb + wrow_off[a] + ((seg ^ wswz[a]) << 4)));
#pragma unroll
            for (int b = 0; b < FF; ++b) afr[b] = as_vec8<P>(*(const uint4*)(Ab + arow_off[b] + ((seg ^ aswz[b]) << 4)));
            if constexpr ((OPT & 4) != 0) {
#pragma unroll
                for (int a = 0; a < FC; ++a) asm volatile("" :: "v"(wfr[a]));
#pragma unroll
                for (int b = 0; b < FF; ++b) asm volatile("" :: "v"(afr[b]));
            } else {
                if constexpr ((OPT & 1) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int a = 0; a < FC; ++a)
#pragma unroll
                    for (int b = 0; b < FF; ++b) acc[a][b] = P::mfma(wfr[a], afr[b], acc[a][b]);
                if constexpr ((OPT & 1) != 0) __builtin_amdgcn_s_setprio(0);
            }
        }
    };

    issueA(0, 0); issueW(0, 0, 0);
    ST_DMA_WAIT(0);
    __syncthreads();
    int it = 0;
    for (int c = 0; c < nch; ++c) {
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            const bool last = (c == nch - 1) && (j == TAPS - 1);
            if constexpr ((OPT & 2) == 0) {
                if ((j == 0) && (c + 1 < nch)) issueA(c + 1, (c + 1) & 1);
                if (!last) { if (j == TAPS - 1) issueW(c + 1, 0, (it + 1) & 1); else issueW(c, j + 1, (it + 1) & 1); }
            }
            compute(c & 1, it & 1, j);
            ST_DMA_WAIT(0);       // the asm-issued LDS-DMA is invisible to hipcc: drain it by hand ...
            __syncthreads();      // ... then fence the buffer swap
            ++it;
        }
    }
    conv_epilogue<P, EPI>(acc, g, n, t0, cbase, wc, wf, l31, hi);
}

template <class P, int TAPS, int EPI, int OPT = 0>
static hipError_t launch_glds(const ConvGemmArgs& a, hipStream_t s) {
    constexpr int AROWS = kBF + TAPS - 1;
    constexpr int lds = ((OPT & 8) ? 1 : 2) * AROWS * 128 + 2 * kBC * 128;
    // the >64 KB dynamic-LDS opt-in is per device: remember it per device id (engines may live on several GPUs)
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    bool& attr_done = attr_done_dev[dev_];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<P, TAPS, EPI, OPT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (!a.zeros) return hipErrorInvalidValue;
    const int total = a.n_items * a.tiles_f * a.tiles_c;
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL((conv_gemm_glds_kernel<P, TAPS, EPI, OPT>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <class P, int TAPS, int EPI, int VAR>
static hipError_t launch_var(const ConvGemmArgs& a, hipStream_t s) {
    constexpr int AROWS = kBF + TAPS - 1;
    constexpr int lds = 2 * AROWS * kLdsRowBytes + 2 * kBC * kLdsRowBytes;
    // the >64 KB dynamic-LDS opt-in is per device: remember it per device id (engines may live on several GPUs)
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    bool& attr_done = attr_done_dev[dev_];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_kernel<P, TAPS, EPI, VAR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int total = a.n_items * a.tiles_f * a.tiles_c;
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL((conv_gemm_kernel<P, TAPS, EPI, VAR>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

// experiment hook: ST_GEMM_VARIANT=0 selects the register-staged K loop instead of the LDS-DMA one
inline int gemm_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ST_GEMM_VARIANT"); v = e ? atoi(e) : 10; }
    return v;
}

template <class P, int TAPS, int EPI>
static hipError_t launch_one(const ConvGemmArgs& a, hipStream_t s) {
    if (((a.c0 + a.c1) & 63) != 0 || (a.cout % kBC) != 0) return hipErrorInvalidValue;
    return gemm_variant() == 0 ? launch_var<P, TAPS, EPI, 0>(a, s) : launch_glds<P, TAPS, EPI>(a, s);
}

template <class P>
static hipError_t launch_conv_gemm_t(int taps, int epi, const ConvGemmArgs& a, hipStream_t s) {
    if (taps == 3 && epi == EPI_ACT16) return launch_one<P, 3, EPI_ACT16>(a, s);
    if (taps == 3 && epi == EPI_F32) return launch_one<P, 3, EPI_F32>(a, s);
    if (taps == 3 && epi == EPI_RESGATE) return launch_one<P, 3, EPI_RESGATE>(a, s);
    if (taps == 1 && epi == EPI_F32) return launch_one<P, 1, EPI_F32>(a, s);
    if (taps == 1 && epi == EPI_RESGATE) return launch_one<P, 1, EPI_RESGATE>(a, s);
    if (taps == 1 && epi == EPI_QKV) return launch_one<P, 1, EPI_QKV>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace st
