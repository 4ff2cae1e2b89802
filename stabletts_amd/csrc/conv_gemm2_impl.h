// Implicit-GEMM Conv1d (k = 1 or 3) for gfx950: LDS-DMA staged K loop + LDS-staged, fully coalesced
// epilogues that can carry the NEXT LayerNorm.
//
// Tile shapes (template): BC output channels x BF frames, WC x WF waves, wave tile (BC/WC) x (BF/WF)
// built from 32x32x16 MFMA fragments.  Shipping configurations:
//     BIG  : 256 ch x 256 fr, 8 waves (2x4) of 128x64     -- 1 block/CU, least LDS / DMA bytes per MFMA; used
//            whenever Cout % 256 == 0 and the launch has >= ~3/4 block per CU
//     T128 : 128 ch x 128 fr, 4 waves (2x2) of 64x64      -- 2 blocks/CU (small grids, Cout = 128)
//     RC   : 256 ch x 128 fr, 8 waves (4x2) of 64x64      -- "row complete" for LayerNorm-carrying epilogues
//            on small grids
//     (+ conv_gemm3_kernel: 128 x 126, three weight buffers, counted vmcnt, deep k=3 convs on small grids)
// A block that owns all 256 hidden channels of its frames can apply FiLM + LayerNorm + adaLN modulate -- the
// prologue of the next op in the reference (estimator.py:16, diffusion_transformer.py:111-112) -- in its
// epilogue and write the 16-bit MFMA operand of the next GEMM directly.
// Epilogues (one per EPI, all ending in full-row coalesced stores):
//     EPI_F32 / EPI_RESGATE : accumulators parked in LDS as [frame][channel] fp32 (conflict-free 16-B stores, row
//            pitch BC+4), rows finished one frame per wave with lane = 4 channels, in batches of 4 rows, phase by
//            phase (g2_rows): bias / mask / gate / residual / FiLM / LayerNorm on contiguous 1 KB rows
//     EPI_ACT16 : SiLU + mask in the accumulator registers, one 16-bit LDS image, 1-KiB row stores (g2_epilogue_act16)
//     EPI_QKV   : RoPE + q scaling in registers, one 16-bit LDS image per q / k / v plane (g2_epilogue_qkv)
#pragma once
#include "common.h"
#include "launch.h"
#include <algorithm>
#include <cstdlib>

#ifndef ST_STAGE_TIMING
#define ST_STAGE_TIMING 0      // 1: s_memtime anatomy of the K loop into ConvGemmArgs::dbg (tools/gemm2_bench.hip only)
#endif

namespace st {

// WBUF = 1: ONE weight buffer (the next stage's weights are requested only after every wave has finished the current
// one: the DMA round trip is exposed inside the block and hidden by a SECOND resident block instead).  Used for the fused
// q/k/v projection on 256 x 128 tiles: 2 x 16 KB + 32 KB = 64 KB of loop buffers under the 74 KB of its epilogue image, so two
// blocks share a CU and one block's HBM-bound epilogue runs beside the other's K loop.
template <int BC, int BF, int WC, int WF, int TAPS, int WBUF = 2>
struct G2Cfg {
    static constexpr int NW = WC * WF, NT = 64 * NW;
    static constexpr int TC = BC / WC, TF = BF / WF, FC = TC / 32, FF = TF / 32;
    static constexpr int AROWS = BF + TAPS - 1;
    static constexpr int A_BYTES = AROWS * 128, W_BYTES = BC * 128;
    static constexpr int PITCH = BC + 4;                       // fp32 words per staged frame row
    static constexpr int STAGE_BYTES = (NW == 8 ? TF : BF) * PITCH * 4;
    static constexpr int STAGE16_BYTES = BF * (BC * 2 + 16);  // 16-bit [frame][channel] image of the whole tile (EPI_ACT16)
    static constexpr int LOOP_BYTES = 2 * A_BYTES + WBUF * W_BYTES;
    static constexpr int LDS_MAX2 = LOOP_BYTES > STAGE_BYTES ? LOOP_BYTES : STAGE_BYTES;
    static constexpr int LDS_BYTES = LDS_MAX2 > STAGE16_BYTES ? LDS_MAX2 : STAGE16_BYTES;
    static_assert((BC / 8) % NW == 0 && (BF / 8) % NW == 0, "DMA pieces must split evenly over the waves");
};

// ---- per-row epilogue ------------------------------------------------------------------------
// Per-lane constants of a block: everything that depends only on (item, channel) is loaded ONCE, outside
// the frame loop (stores in the loop would otherwise force the compiler to reload them every frame).
struct G2Consts { float4 bias, gate, ga, be, sh, sc; };

template <int EPI, bool LN>
__device__ __forceinline__ G2Consts g2_consts(const ConvGemmArgs& g, int n, int ch) {
    G2Consts k;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    k.bias = g.bias ? *(const float4*)(g.bias + ch) : z;
    k.gate = z; k.ga = make_float4(1.f, 1.f, 1.f, 1.f); k.be = z; k.sh = z; k.sc = z;
    if constexpr (EPI == EPI_RESGATE) k.gate = *(const float4*)(g.gate + (size_t)n * g.gate_stride + ch);
    if constexpr (LN) {
        if (g.ln_h16) {
            if (g.ln_film) {
                const float* f = g.ln_film + (size_t)(n % g.ln_film_mod) * g.ln_film_stride + ch;
                k.ga = *(const float4*)f; k.be = *(const float4*)(f + 256);
            }
            const float* ad = g.ln_ada + (size_t)n * g.ln_ada_stride + ch;
            k.sh = *(const float4*)(ad + g.ln_shift_off); k.sc = *(const float4*)(ad + g.ln_scale_off);
        }
    }
    return k;
}

// Finishes R rows at once: row u = 4 consecutive channels [ch, ch+4) of frame t[u] of item n.  v = accumulator
// values, m = mask value of the frame, xin = preloaded residual row (EPI_RESGATE) or addend row (EPI_F32 with
// add32), ok = row inside the tensor (only the STORES are predicated: the arithmetic of an invalid row runs on
// clamped loads and is dropped).  Written phase by phase over all R rows -- every flag test sits outside a row
// loop -- so each phase is one basic block and hipcc interleaves the rows' dependent chains (LayerNorm
// reductions, exp/rcp) instead of serialising them behind per-row branches.
// LN = the wave holds complete 256-channel rows, so FiLM + LayerNorm + modulate of the next op can be applied.
template <class P, int EPI, bool LN, int R>
__device__ __forceinline__ void g2_rows(const ConvGemmArgs& g, const G2Consts& k, int n, const int (&t)[R],
                                        const bool (&ok)[R], int ch, float4 (&v)[R], const float (&m)[R],
                                        const float4 (&xin)[R]) {
    size_t grow[R];
#pragma unroll
    for (int u = 0; u < R; ++u) grow[u] = (size_t)n * g.T + t[u];
    static_assert(EPI != EPI_ACT16, "EPI_ACT16 has its own epilogue (g2_epilogue_act16)");
    {
        if constexpr (EPI == EPI_F32) {
            const bool msk = g.flags & GF_MASK;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const float mm = msk ? m[u] : 1.0f;
                v[u].x = (v[u].x + k.bias.x + xin[u].x) * mm; v[u].y = (v[u].y + k.bias.y + xin[u].y) * mm;
                v[u].z = (v[u].z + k.bias.z + xin[u].z) * mm; v[u].w = (v[u].w + k.bias.w + xin[u].w) * mm;
            }
        } else {   // EPI_RESGATE: x + gate * ((acc + b) * mask)
            if (g.branch32) {
#pragma unroll
                for (int u = 0; u < R; ++u)
                    if (ok[u]) store_row16(g.branch32 + grow[u] * g.cout + ch,
                                           make_float4((v[u].x + k.bias.x) * m[u], (v[u].y + k.bias.y) * m[u], (v[u].z + k.bias.z) * m[u], (v[u].w + k.bias.w) * m[u]));
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                v[u].x = xin[u].x + k.gate.x * ((v[u].x + k.bias.x) * m[u]); v[u].y = xin[u].y + k.gate.y * ((v[u].y + k.bias.y) * m[u]);
                v[u].z = xin[u].z + k.gate.z * ((v[u].z + k.bias.z) * m[u]); v[u].w = xin[u].w + k.gate.w * ((v[u].w + k.bias.w) * m[u]);
            }
        }
        if (g.out16) {
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (ok[u]) store_row8((unsigned char*)g.out16 + (grow[u] * g.cout + ch) * 2, pack4<P>(v[u].x, v[u].y, v[u].z, v[u].w));
            if (g.out16_lo) {      // split-precision copy: lo = x - float(hi), the operand pair of final_proj
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const float4 l = make_float4(v[u].x - (float)to16<P>(v[u].x), v[u].y - (float)to16<P>(v[u].y),
                                                 v[u].z - (float)to16<P>(v[u].z), v[u].w - (float)to16<P>(v[u].w));
                    if (ok[u]) store_row8((unsigned char*)g.out16_lo + (grow[u] * g.cout + ch) * 2, pack4<P>(l.x, l.y, l.z, l.w));
                }
            }
        }
        if constexpr (LN) {
            if (g.ln_h16) {
                // the next op's prologue: FiLM (estimator.py:31-33,16), LayerNorm, adaLN modulate
                if (g.ln_film) {
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        v[u].x = (k.ga.x * v[u].x + k.be.x) * m[u]; v[u].y = (k.ga.y * v[u].y + k.be.y) * m[u];
                        v[u].z = (k.ga.z * v[u].z + k.be.z) * m[u]; v[u].w = (k.ga.w * v[u].w + k.be.w) * m[u];
                    }
                }
                if (g.out32) {
#pragma unroll
                    for (int u = 0; u < R; ++u)
                        if (ok[u]) store_row16(g.out32 + grow[u] * g.cout + ch, v[u]);
                }
                float mean[R], rstd[R];
#pragma unroll
                for (int u = 0; u < R; ++u) mean[u] = wave_sum(v[u].x + v[u].y + v[u].z + v[u].w) * (1.0f / 256.0f);
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    v[u].x -= mean[u]; v[u].y -= mean[u]; v[u].z -= mean[u]; v[u].w -= mean[u];
                    rstd[u] = wave_sum(v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w) * (1.0f / 256.0f);
                }
                const bool mout = g.ln_mask_out;
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const float rs = 1.0f / sqrtf(rstd[u] + 1e-5f);
                    const float mm = mout ? m[u] : 1.0f;
                    const float h0 = (v[u].x * rs * (1.0f + k.sc.x) + k.sh.x) * mm, h1 = (v[u].y * rs * (1.0f + k.sc.y) + k.sh.y) * mm;
                    const float h2 = (v[u].z * rs * (1.0f + k.sc.z) + k.sh.z) * mm, h3 = (v[u].w * rs * (1.0f + k.sc.w) + k.sh.w) * mm;
                    if (ok[u]) store_row8((unsigned char*)g.ln_h16 + (grow[u] * 256 + ch) * 2, pack4<P>(h0, h1, h2, h3));
                }
                return;
            }
        }
        if (g.out32 && !(EPI == EPI_RESGATE && g.out32_readonly)) {
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (ok[u]) store_row16(g.out32 + grow[u] * g.cout + ch, v[u]);
        }
    }
}

// LDS-staged epilogue shared by the gen-2 kernels.  fvalid = number of valid frame columns of the tile
// (BF, or BF-2 for the 3-buffer k=3 kernel whose activation tile includes its own halo).
// `park(fbase)` stores the calling wave's accumulator tile into `stage` as [frame][channel] fp32, frame rows from fbase on (row
// pitch BC + 4 words): the only part that depends on the accumulator layout of the K loop (g2_epilogue: 32x32x16 fragments).
template <class P, int EPI, int BC, int BF, int WC, int WF, class Park>
__device__ __forceinline__ void g2_epilogue_core(Park park, float* stage, const ConvGemmArgs& g,
                                                 int n, int t0, int fvalid, int cbase, int wave, int lane) {
    constexpr int NW = WC * WF, TF = BF / WF, PITCH = BC + 4;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wf = wave / WC;
    const int T = g.T;
    static_assert(EPI != EPI_QKV && EPI != EPI_ACT16 && EPI != EPI_SILU, "QKV: g2_epilogue_qkv; ACT16: g2_epilogue_act16; SILU: g2_epilogue_silu");
    static_assert(BC == 128 || BC == 256, "row walker handles 128 or 256 channels");
    constexpr bool LN = (BC == 256);
    // frames staged per pass: the 4-wave 128x128 tile goes in ONE pass (fewer barriers, one exposed global-load
    // latency: out_proj 38 -> 35 us); the 8-wave tiles keep 64-frame passes (128-frame passes need 16 preloaded
    // residual rows per lane on top of the live accumulators and spill: FFN conv_2 104 -> 113 us)
    constexpr int PASSF = (NW == 8) ? TF : BF, WFP = PASSF / TF, NPASS = BF / PASSF;
    static_assert(PASSF % TF == 0 && BF % PASSF == 0, "pass geometry");
    constexpr int RPW = LN ? PASSF / NW : PASSF / (2 * NW);  // wave-rows (1 or 2 frames each) per wave per pass
    const int chl = LN ? lane * 4 : l31 * 4;                 // this lane's channel quad inside the block tile
    const G2Consts kc = g2_consts<EPI, LN>(g, n, cbase + chl);
    const float* mrow = g.mask ? g.mask + (size_t)(n % g.mask_mod) * T : nullptr;
    const int an = n < g.add_clamp ? n : g.add_clamp;
    // ragged batches: rows of this tile at or past the item's last needed frame (t_lim, launch.h) are neither stored nor fetched --
    // their loads are clamped to the last needed row (one cached line) -- so the HBM-bound row phase shrinks with the padding
    // at ROW granularity, not only by whole tiles
    const int tlim = g.t_lim ? min(T, g.t_lim[n % g.t_lim_mod]) : T;
#pragma unroll 1
    for (int p = 0; p < NPASS; ++p) {
        if (wf / WFP == p) park((wf % WFP) * TF);
        const int tbase = t0 + p * PASSF;
        // every global input of this wave's rows is requested before the barrier: the latency overlaps the other
        // waves' staging stores and the barrier wait
        float mk[RPW]; float4 xin[RPW]; int fr[RPW], tt[RPW]; bool ok[RPW];
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int f = LN ? (wave + u * NW) : ((wave + u * NW) * 2 + hi);
            const int t = tbase + f;
            fr[u] = f; tt[u] = t;
            ok[u] = (t < tlim) && (p * PASSF + f < fvalid);
            const int tl = t < tlim ? t : tlim - 1;      // loads of an invalid row are clamped, its stores dropped
            mk[u] = mrow ? mrow[tl] : 1.0f;
            xin[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (EPI == EPI_RESGATE) xin[u] = *(const float4*)((g.res32 ? g.res32 : g.out32) + ((size_t)n * T + tl) * g.cout + cbase + chl);
            if constexpr (EPI == EPI_F32) { if (g.add32) xin[u] = *(const float4*)(g.add32 + ((size_t)an * T + tl) * g.cout + cbase + chl); }
        }
        __syncthreads();
        // ... then the rows are finished from LDS, all RPW of them phase by phase
        // (sub-batches of RB rows: 8 rows of the fp32 epilogues in flight on top of the live accumulators spill)
        constexpr int RB = (EPI != EPI_ACT16 && RPW > 4) ? 4 : RPW;
#pragma unroll
        for (int b0 = 0; b0 < RPW; b0 += RB) {
            float mk2[RB]; float4 xin2[RB], v[RB]; int tt2[RB]; bool ok2[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                v[u] = *(const float4*)(stage + fr[b0 + u] * PITCH + chl);
                mk2[u] = mk[b0 + u]; xin2[u] = xin[b0 + u]; tt2[u] = tt[b0 + u]; ok2[u] = ok[b0 + u];
            }
            g2_rows<P, EPI, LN, RB>(g, kc, n, tt2, ok2, cbase + chl, v, mk2, xin2);
        }
        if (p + 1 < NPASS) __syncthreads();
    }
}

template <class P, int EPI, int BC, int BF, int WC, int WF>
__device__ __forceinline__ void g2_epilogue(f32x16_t (&acc)[BC / WC / 32][BF / WF / 32], float* stage, const ConvGemmArgs& g,
                                            int n, int t0, int fvalid, int cbase, int wave, int lane) {
    constexpr int TC = BC / WC, FC = TC / 32, FF = BF / WF / 32, PITCH = BC + 4;
    const int l31 = lane & 31, hi = lane >> 5, wc = wave % WC;
    g2_epilogue_core<P, EPI, BC, BF, WC, WF>([&](int fbase) {
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int b = 0; b < FF; ++b)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int fl = fbase + b * 32 + l31;
                    const int ch = wc * TC + a * 32 + 8 * q4 + 4 * hi;
                    *(float4*)(stage + fl * PITCH + ch) = make_float4(acc[a][b][4 * q4 + 0], acc[a][b][4 * q4 + 1],
                                                                      acc[a][b][4 * q4 + 2], acc[a][b][4 * q4 + 3]);
                }
    }, stage, g, n, t0, fvalid, cbase, wave, lane);
}

// Accumulator start value: EPI_ACT16 / EPI_QKV kernels start from the bias (one add per output saved in the epilogue).
template <int EPI, int FC, int FF>
__device__ __forceinline__ void g2_init_acc(f32x16_t (&acc)[FC][FF], const ConvGemmArgs& g, int chw, int hi) {
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (EPI == EPI_ACT16 || EPI == EPI_GELU16 || EPI == EPI_QKV) { if (g.bias) bv = *(const float4*)(g.bias + chw + a * 32 + 8 * q4 + 4 * hi); }
#pragma unroll
            for (int b = 0; b < FF; ++b) {
                acc[a][b][4 * q4 + 0] = bv.x; acc[a][b][4 * q4 + 1] = bv.y;
                acc[a][b][4 * q4 + 2] = bv.z; acc[a][b][4 * q4 + 3] = bv.w;
            }
        }
}

// EPI_ACT16 epilogue (activation tensors that only feed the next GEMM: FFN conv_1, prenet convs).  SiLU and
// the frame mask are applied IN the accumulator registers (lane = frame, so the mask is one value per fragment
// column; every wave works at once, no barrier, no per-row control flow), the results are packed to 16 bit and
// parked in LDS as [frame][channel] (row pitch BC*2+16 B), and after ONE barrier the block writes the tile as
// whole rows: 16 B per lane, 64 lanes = 1 KiB of consecutive HBM bytes per store instruction.
template <class P, int BC, int BF, int WC, int WF, bool GELU = false>
__device__ __forceinline__ void g2_epilogue_act16(f32x16_t (&acc)[BC / WC / 32][BF / WF / 32], unsigned char* stage,
                                                  const ConvGemmArgs& g, int n, int t0, int fvalid, int cbase, int wave, int lane) {
    constexpr int NW = WC * WF, TC = BC / WC, TF = BF / WF, FC = TC / 32, FF = TF / 32, PB = BC * 2 + 16;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wave % WC, wf = wave / WC;
    const int T = g.T;
    const bool do_silu = g.flags & GF_SILU;
    const float* mrow = ((g.flags & GF_MASK) && g.mask) ? g.mask + (size_t)(n % g.mask_mod) * T : nullptr;
#pragma unroll
    for (int b = 0; b < FF; ++b) {
        const int fl = wf * TF + b * 32 + l31;
        const int t = t0 + fl;
        const float m = mrow ? mrow[t < T ? t : T - 1] : 1.0f;
        const bool allone = __all(m == 1.0f);
#pragma unroll
        for (int a = 0; a < FC; ++a) {          // one 32x32 fragment at a time: its registers die at the ds_write
            f32x16_t v = acc[a][b];
            if constexpr (GELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
            } else if (do_silu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = silu_fast(v[r]);
            }
            if (!allone) {                      // wave-uniform: tiles inside the valid prefix skip the multiply
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= m;
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int ch = wc * TC + a * 32 + 8 * q4 + 4 * hi;
                *(uint2*)(stage + fl * PB + ch * 2) = pack4<P>(v[4 * q4 + 0], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
            }
        }
    }
    __syncthreads();
    constexpr int LPR = BC / 8, RPI = 64 / LPR;        // lanes per row, rows per wave instruction
    const int rsub = lane / LPR, cl = lane % LPR;
    unsigned char* obase = (unsigned char*)g.out16 + (((size_t)n * T + t0) * g.cout + cbase) * 2 + cl * 16;
#pragma unroll
    for (int i = 0; i < BF / (NW * RPI); ++i) {
        const int f = (i * NW + wave) * RPI + rsub;
        const uint4 v = *(const uint4*)(stage + f * PB + cl * 16);
        if (t0 + f < T && f < fvalid) store_row16(obase + (size_t)f * g.cout * 2, v);
    }
}

// EPI_SILU epilogue (training FFN, diffusion_transformer.py:25-30): the element-wise step between conv_1 and conv_2 -- SiLU,
// dropout, frame mask -- applied in the accumulator registers like EPI_ACT16, every global access a full coalesced row
// through the 16-bit LDS image [frame][channel].
//   forward  (g.act16):  image 1 = round16(acc + bias) -> out16 (the pre-activation the backward needs);
//                        image 2 = silu(float(image 1)) * dropout factor * mask -> act16 (conv_2's operand).
//   backward (g.dact16): the pre-activation tile is loaded row-wise into the image, every lane reads its own
//                        (frame, 4 channels) groups back, out16 = acc * mask * dropout factor * silu'(pre) in place.
// Same values as silu_drop_kernel / silu_bwd_kernel (train_kernels.hip) compute from the same rounded inputs.
template <class P, int BC, int BF, int WC, int WF>
__device__ __forceinline__ void g2_epilogue_silu(f32x16_t (&acc)[BC / WC / 32][BF / WF / 32], unsigned char* stage,
                                                 const ConvGemmArgs& g, int n, int t0, int fvalid, int cbase, int wave, int lane) {
    constexpr int NW = WC * WF, TC = BC / WC, TF = BF / WF, FC = TC / 32, FF = TF / 32, PB = BC * 2 + 16;
    constexpr int LPR = BC / 8, RPI = 64 / LPR;        // lanes per row, rows per wave instruction
    typedef __attribute__((ext_vector_type(4))) typename P::elem v4;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wave % WC, wf = wave / WC;
    const int T = g.T;
    const float* mrow = g.mask ? g.mask + (size_t)(n % g.mask_mod) * T : nullptr;
    struct { unsigned long long seed; unsigned thresh16; float scale; } d = {g.drop_seed, g.drop_thresh16, g.drop_scale};
    const int rsub = lane / LPR, cl = lane % LPR;
    const size_t tile0 = (((size_t)n * T + t0) * g.cout + cbase) * 2 + cl * 16;
    auto rows_out = [&](void* dst) {
#pragma unroll
        for (int i = 0; i < BF / (NW * RPI); ++i) {
            const int f = (i * NW + wave) * RPI + rsub;
            const uint4 v = *(const uint4*)(stage + f * PB + cl * 16);
            if (t0 + f < T && f < fvalid) store_row16((unsigned char*)dst + tile0 + (size_t)f * g.cout * 2, v);
        }
    };
    const bool fwd = g.act16 != nullptr;
    if (!fwd) {     // pre-activation tile -> image (rows outside the tensor: zeros, their results are never stored)
#pragma unroll
        for (int i = 0; i < BF / (NW * RPI); ++i) {
            const int f = (i * NW + wave) * RPI + rsub;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (t0 + f < T && f < fvalid) v = *(const uint4*)((const unsigned char*)g.dact16 + tile0 + (size_t)f * g.cout * 2);
            *(uint4*)(stage + f * PB + cl * 16) = v;
        }
        __syncthreads();
    }
    // element (fragment a, b; register group q4): frame fl = wf*TF + b*32 + l31, channels ch .. ch+3, ch = wc*TC + a*32 + 8*q4 + 4*hi
    auto factors = [&](int t, int ch, float2& f01, float2& f23) {
        f01 = make_float2(1.0f, 1.0f); f23 = f01;
        if (d.thresh16) {       // FFN element-pair hash of element index row * cout + channel (launch.h: DropCfg)
            const unsigned long long i = ((unsigned long long)n * T + t) * g.cout + cbase + ch;
            f01 = drop_factors2(d, drop_ffn_hash(d, i)); f23 = drop_factors2(d, drop_ffn_hash(d, i + 2));
        }
    };
    if (fwd) {
#pragma unroll
        for (int b = 0; b < FF; ++b) {
            const int fl = wf * TF + b * 32 + l31;
#pragma unroll
            for (int a = 0; a < FC; ++a)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int ch = wc * TC + a * 32 + 8 * q4 + 4 * hi;
                    const float4 bv = g.bias ? *(const float4*)(g.bias + cbase + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *(uint2*)(stage + fl * PB + ch * 2) = pack4<P>(acc[a][b][4 * q4 + 0] + bv.x, acc[a][b][4 * q4 + 1] + bv.y,
                                                                   acc[a][b][4 * q4 + 2] + bv.z, acc[a][b][4 * q4 + 3] + bv.w);
                }
        }
        __syncthreads();
        rows_out(g.out16);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < FF; ++b) {
        const int fl = wf * TF + b * 32 + l31;
        const int t = t0 + fl;
        const float m = mrow ? mrow[t < T ? t : T - 1] : 1.0f;
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int ch = wc * TC + a * 32 + 8 * q4 + 4 * hi;
                float2 f01, f23;
                factors(t, ch, f01, f23);
                const v4 pre = *(const v4*)(stage + fl * PB + ch * 2);
                uint2 o;
                if (fwd)
                    o = pack4<P>(silu_fast((float)pre[0]) * f01.x * m, silu_fast((float)pre[1]) * f01.y * m,
                                 silu_fast((float)pre[2]) * f23.x * m, silu_fast((float)pre[3]) * f23.y * m);
                else
                    o = pack4<P>(acc[a][b][4 * q4 + 0] * m * f01.x * silu_grad_fast((float)pre[0]), acc[a][b][4 * q4 + 1] * m * f01.y * silu_grad_fast((float)pre[1]),
                                 acc[a][b][4 * q4 + 2] * m * f23.x * silu_grad_fast((float)pre[2]), acc[a][b][4 * q4 + 3] * m * f23.y * silu_grad_fast((float)pre[3]));
                *(uint2*)(stage + fl * PB + ch * 2) = o;
            }
    }
    __syncthreads();
    rows_out(fwd ? g.act16 : g.out16);
}

// EPI_QKV epilogue (fused q/k/v projection of diffusion_transformer.py:60-62, cout = 3 x 256, tile = 256 channels
// = exactly the q, the k or the v plane of 256 frames).  Like EPI_ACT16 everything elementwise happens in the
// accumulator registers and the tile goes through ONE 16-bit LDS image so that every global store is a full
// coalesced row:
//   q / k : partial RoPE (pairs d, d+16 for d < 16 are register groups q4, q4+2 of the head's first fragment:
//           lane local; diffusion_transformer.py:180-198), q scaled by log2(e)/sqrt(64); image [head][frame][64]
//           (pitch 144 B) -> q/k [item][H][T][64]: the 256 frames of a head are one contiguous 32 KB run.
//   v     : image [channel][frame] with the PV-operand key order (bits 2<->3 of the frame index swapped inside
//           every 16, attention.hip), frames >= T zeroed -> vT [item][H][64][Tp]: 512 B runs per (head, dim).
constexpr int kQkvRowPitch = 144;
template <int BC, int BF> constexpr int g2_qkv_lds_bytes() {
    return (BC / 64) * BF * kQkvRowPitch > BC * (BF * 2 + 16) ? (BC / 64) * BF * kQkvRowPitch : BC * (BF * 2 + 16);
}
template <class P, int BC, int BF, int WC, int WF>
__device__ __forceinline__ void g2_epilogue_qkv(f32x16_t (&acc)[BC / WC / 32][BF / WF / 32], unsigned char* stage,
                                                const ConvGemmArgs& g, int n, int t0, int cbase, int wave, int lane) {
    static_assert(BC == 256 && (BC / WC) % 64 == 0, "one q/k/v plane per block, whole heads per wave");
    constexpr int NW = WC * WF, TC = BC / WC, TF = BF / WF, FF = TF / 32, HPW = TC / 64, PQ = kQkvRowPitch, PV = BF * 2 + 16;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wave % WC, wf = wave / WC;
    const int T = g.T, H = g.n_heads;
    const int which = cbase / BC;                 // 0 q, 1 k, 2 v
    const int tlim = g.t_lim ? min(T, g.t_lim[n % g.t_lim_mod]) : T;      // ragged batches: q / k rows past the last needed frame are not stored
    if (which < 2) {
        const float sc = which == 0 ? g.qscale : 1.0f;
        unsigned char* dst_lo = (unsigned char*)(which == 0 ? g.q_lo : g.k_lo);
        // pass 0: the 16-bit plane; pass 1 (split-precision attention operands only): the rounding residuals of the same values
        // through the same LDS image -- the RoPE rotation is recomputed (same instructions: the hi part is bit-identical)
        for (int pass = 0; pass < (dst_lo ? 2 : 1); ++pass) {
        if (pass) __syncthreads();      // the row stores of pass 0 have read the image
#pragma unroll
        for (int b = 0; b < FF; ++b) {
            const int fl = wf * TF + b * 32 + l31;
            const int tl = t0 + fl < T ? t0 + fl : T - 1;
            float cc[2][4], ss[2][4];
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const float4 cs = *(const float4*)(g.rope_cos + (size_t)tl * 16 + 8 * q4 + 4 * hi);
                const float4 sn = *(const float4*)(g.rope_sin + (size_t)tl * 16 + 8 * q4 + 4 * hi);
                cc[q4][0] = cs.x; cc[q4][1] = cs.y; cc[q4][2] = cs.z; cc[q4][3] = cs.w;
                ss[q4][0] = sn.x; ss[q4][1] = sn.y; ss[q4][2] = sn.z; ss[q4][3] = sn.w;
            }
#pragma unroll
            for (int hh = 0; hh < HPW; ++hh) {
                f32x16_t r = acc[2 * hh][b];
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x1 = r[4 * q4 + e], x2 = r[4 * (q4 + 2) + e];
                        rope_rot(x1, x2, cc[q4][e], ss[q4][e]);
                        r[4 * q4 + e] = x1; r[4 * (q4 + 2) + e] = x2;
                    }
                unsigned char* row = stage + ((wc * HPW + hh) * BF + fl) * PQ + 4 * hi * 2;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    if (pass == 0) {
                        *(uint2*)(row + 16 * q4) = scale_pack4<P>(r[4 * q4 + 0], r[4 * q4 + 1], r[4 * q4 + 2], r[4 * q4 + 3], sc);
                        *(uint2*)(row + 64 + 16 * q4) = scale_pack4<P>(acc[2 * hh + 1][b][4 * q4 + 0], acc[2 * hh + 1][b][4 * q4 + 1],
                                                                       acc[2 * hh + 1][b][4 * q4 + 2], acc[2 * hh + 1][b][4 * q4 + 3], sc);
                    } else {
                        *(uint2*)(row + 16 * q4) = scale_pack4_lo<P>(r[4 * q4 + 0], r[4 * q4 + 1], r[4 * q4 + 2], r[4 * q4 + 3], sc);
                        *(uint2*)(row + 64 + 16 * q4) = scale_pack4_lo<P>(acc[2 * hh + 1][b][4 * q4 + 0], acc[2 * hh + 1][b][4 * q4 + 1],
                                                                          acc[2 * hh + 1][b][4 * q4 + 2], acc[2 * hh + 1][b][4 * q4 + 3], sc);
                    }
                }
            }
        }
        __syncthreads();
        unsigned char* dst = pass ? dst_lo : (unsigned char*)(which == 0 ? g.q : g.k);
        const int rsub = lane >> 3, seg = lane & 7;
#pragma unroll
        for (int i = 0; i < (BC / 64) * BF / (NW * 8); ++i) {
            const int rowid = (i * NW + wave) * 8 + rsub;
            const int head = rowid / BF, f = rowid % BF;
            const uint4 v = *(const uint4*)(stage + rowid * PQ + seg * 16);
            if (t0 + f < tlim) store_row16(dst + (((size_t)n * H + head) * T + t0 + f) * 128 + seg * 16, v);
        }
        }
    } else {
        // pass 0: the 16-bit plane; pass 1 (training with g.vt_lo): the rounding residuals v - float(v16) through the same image
        auto plane = [&](auto pass_c, unsigned char* dst) {
            constexpr bool LO = decltype(pass_c)::value;
#pragma unroll
            for (int b = 0; b < FF; ++b) {
                const int fl = wf * TF + b * 32 + l31;
                const bool tv = t0 + fl < T;
                const int pos = (fl & ~12) | ((fl & 4) << 1) | ((fl & 8) >> 1);
#pragma unroll
                for (int a = 0; a < TC / 32; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = wc * TC + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        float x = tv ? acc[a][b][r] : 0.0f;
                        if constexpr (LO) {
                            asm volatile("" : "+v"(x));      // (keeps the 128 residuals from being formed ahead of the first plane's stores: registers)
                            x -= (float)to16<P>(x);
                        }
                        *(typename P::elem*)(stage + ch * PV + pos * 2) = to16<P>(x);
                    }
            }
            __syncthreads();
            constexpr int SPR = BF / 8, RPI = 64 / SPR;      // 16-B segments per channel row, channel rows per wave instruction
            const int rsub = lane / SPR, seg = lane % SPR;
#pragma unroll
            for (int i = 0; i < BC / (NW * RPI); ++i) {
                const int ch = (i * NW + wave) * RPI + rsub;
                const uint4 v = *(const uint4*)(stage + ch * PV + seg * 16);
                const int tcol = t0 + seg * 8;
                if (tcol < g.Tp)
                    store_row16(dst + ((((size_t)n * H + (ch >> 6)) * 64 + (ch & 63)) * g.Tp + tcol) * 2, v);
            }
        };
        plane(std::false_type{}, (unsigned char*)g.vt);
        if (g.vt_lo) {
            __syncthreads();      // the row stores of the first plane have read the image
            plane(std::true_type{}, (unsigned char*)g.vt_lo);
        }
    }
}

template <class P, int TAPS, int EPI, int BC, int BF, int WC, int WF, int WBUF = 2>
__global__ __launch_bounds__(64 * WC * WF, WBUF == 1 ? 4 : 2)          // WBUF = 1 exists to keep two 8-wave blocks on a CU: <= 128 VGPRs
void conv_gemm2_kernel(const ConvGemmArgs g) {
    using vec8 = typename P::vec8;
    using K = G2Cfg<BC, BF, WC, WF, TAPS, WBUF>;
    constexpr int NW = K::NW, FC = K::FC, FF = K::FF, TC = K::TC, TF = K::TF;
    constexpr int A_BYTES = K::A_BYTES, W_BYTES = K::W_BYTES;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Ws = smem + 2 * A_BYTES;

    // split-K launches (g.ksplit > 1, small grids): `ksplit` consecutive blocks share an output tile and each
    // contracts its own range of 64-channel chunks; the raw fp32 partial sums go to plane kz of out32
    // ([ksplit][items][T][cout]) and splitk_finish_kernel applies the epilogue to their sum.
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    const int total = g.n_items * g.tiles_f * g.tiles_c * ks;
    const int per_xcd = gridDim.x >> 3;
    const int lin0 = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin0 >= total) return;
    const int kz = lin0 % ks;
    const int lin = lin0 / ks;
    const int tc = lin % g.tiles_c;
    const int rest = lin / g.tiles_c;
    const int tf = rest % g.tiles_f;
    const int n = rest / g.tiles_f;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wave % WC, wf = wave / WC;
    const int cbase = tc * BC, t0 = tf * BF;
    if (g.t_lim && t0 >= g.t_lim[n % g.t_lim_mod]) return;      // ragged batch: this tile lies past the item's last needed frame
    const int cin = g.c0 + g.c1 + g.c2;
    // (EPI_QKV with GF_K2_V_ONLY: the q and k planes' weights are zero beyond c0 -- their blocks skip those chunks, same sums)
    const int nch_all = (EPI == EPI_QKV && (g.flags & GF_K2_V_ONLY) && cbase < 2 * BC ? g.c0 : cin) >> 6;
    const int cb = kz * nch_all / ks, nch = (kz + 1) * nch_all / ks;      // this block's chunks [cb, nch)
    const int T = g.T;

    const unsigned char* a0 = (const unsigned char*)g.a0 + (size_t)(n % g.a0_mod) * T * g.c0 * 2;
    const unsigned char* a1 = g.c1 ? (const unsigned char*)g.a1 + (size_t)(n % g.a1_mod) * T * g.c1 * 2 : nullptr;
    const unsigned char* wsrc = (const unsigned char*)g.w + (size_t)n * g.w_item_stride;

    // LDS-DMA addressing.  The loop-invariant per-lane part of every source address is computed once (the
    // per-stage part is scalar arithmetic on an SGPR base): issuing a 1-KiB piece costs the issuing wave ~100
    // cycles, address VALU included.  Activation rows outside [0, T) are never transferred; their LDS rows are
    // zeroed once, here (both buffers).
    constexpr int WPW = (BC / 8) / NW, APW = (BF / 8) / NW;       // 1-KiB DMA pieces per wave per tile
    const int prow = lane >> 3;
    unsigned voffW[WPW], voffA0[APW + 1], voffA1[APW + 1];
    bool validA[APW + 1];
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
        const int row = (wave * WPW + k) * 8 + prow;
        voffW[k] = (unsigned)((cbase + row) * TAPS * cin * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int k = 0; k <= APW; ++k) {
        const int row = (k < APW) ? (wave * APW + k) * 8 + prow : BF + prow;     // k == APW: the two halo rows (k = 3)
        const int t = t0 + row - (TAPS / 2);
        const unsigned segb = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        const bool mine = (k < APW) || (TAPS == 3 && wave == 0 && lane < 16);
        validA[k] = mine && (t >= 0 && t < T);
        voffA0[k] = (unsigned)(t * g.c0 * 2) + segb;
        voffA1[k] = (unsigned)(t * g.c1 * 2) + segb;
        if (mine && !(t >= 0 && t < T)) {
            const int pc = (k < APW) ? wave * APW + k : BF / 8;
            *(uint4*)(As + pc * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
            *(uint4*)(As + A_BYTES + pc * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    auto issueW = [&](int c, int j, int buf) {
        const unsigned char* sb = wsrc + (size_t)(j * cin + (c << 6)) * 2;
#pragma unroll
        for (int k = 0; k < WPW; ++k) glds16s(sb, voffW[k], Ws + buf * W_BYTES + (wave * WPW + k) * 1024);
    };
    // K source of channel chunk c: [0, c0) -> a0, [c0, c0 + c1) -> a1 (torch.cat without the copy), and
    // [c0 + c1, c0 + c1 + c2) -> a0 AGAIN from its channel 0 (split-precision operands: [x_hi | x_lo | x_hi]
    // against packed weights [W_hi | W_hi | W_lo])
    auto issueA = [&](int c, int buf) {
        int ch0 = c << 6;
        if (ch0 >= g.c0 + g.c1) ch0 -= g.c0 + g.c1;
        // (two copies of the unrolled piece loop under a wave-uniform branch: a per-piece `first ? voffA0[k] : voffA1[k]`
        // makes hipcc index the two arrays dynamically and park them in scratch -- 80 B/lane, -20 % on every conv)
        if (ch0 < g.c0) {
            const unsigned char* sb = a0 + (size_t)ch0 * 2;
#pragma unroll
            for (int k = 0; k <= APW; ++k) {
                if (k == APW && TAPS != 3) continue;
                const int pc = (k < APW) ? wave * APW + k : BF / 8;
                if (validA[k]) glds16s(sb, voffA0[k], As + buf * A_BYTES + pc * 1024);
            }
        } else {
            const unsigned char* sb = a1 + (size_t)(ch0 - g.c0) * 2;
#pragma unroll
            for (int k = 0; k <= APW; ++k) {
                if (k == APW && TAPS != 3) continue;
                const int pc = (k < APW) ? wave * APW + k : BF / 8;
                if (validA[k]) glds16s(sb, voffA1[k], As + buf * A_BYTES + pc * 1024);
            }
        }
    };

    f32x16_t acc[FC][FF];
    g2_init_acc<EPI, FC, FF>(acc, g, cbase + wc * TC, hi);

    int wrow_off[FC], wswz[FC];
#pragma unroll
    for (int a = 0; a < FC; ++a) {
        const int row = wc * TC + a * 32 + l31;
        wrow_off[a] = row * 128; wswz[a] = (row >> 1) & 7;
    }
    auto compute = [&](int abuf, int wbuf, int j) {
        const unsigned char* Ab = As + abuf * A_BYTES;
        const unsigned char* Wb = Ws + wbuf * W_BYTES;
        int arow_off[FF], aswz[FF];
#pragma unroll
        for (int b = 0; b < FF; ++b) {
            const int row = wf * TF + b * 32 + l31 + j;
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
#pragma unroll
            for (int a = 0; a < FC; ++a)
#pragma unroll
                for (int b = 0; b < FF; ++b) acc[a][b] = P::mfma(wfr[a], afr[b], acc[a][b]);
        }
    };

    issueA(cb, cb & 1); issueW(cb, 0, 0);
    ST_DMA_WAIT(0);
    __syncthreads();
#if ST_STAGE_TIMING
    unsigned long long tm_issue = 0, tm_compute = 0, tm_dma = 0, tm_barrier = 0, tm_n = 0;
    unsigned long long tS = __builtin_amdgcn_s_memtime();
    const unsigned long long tStart = tS;
#endif
    int it = 0;
    for (int c = cb; c < nch; ++c) {
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            const bool last = (c == nch - 1) && (j == TAPS - 1);
            if constexpr (WBUF == 1) {
                if ((j == 0) && (c + 1 < nch)) issueA(c + 1, (c + 1) & 1);
                compute(c & 1, 0, j);
                __syncthreads();                // every wave is done with the one weight buffer ...
                if (!last) { if (j == TAPS - 1) issueW(c + 1, 0, 0); else issueW(c, j + 1, 0); }
                ST_DMA_WAIT(0);                 // ... its refill (and the next activation chunk) has landed
                __syncthreads();
                ++it;
                continue;
            }
            if ((j == 0) && (c + 1 < nch)) issueA(c + 1, (c + 1) & 1);
            if (!last) { if (j == TAPS - 1) issueW(c + 1, 0, (it + 1) & 1); else issueW(c, j + 1, (it + 1) & 1); }
#if ST_STAGE_TIMING
            const unsigned long long tA = __builtin_amdgcn_s_memtime();
#endif
            compute(c & 1, it & 1, j);      // the DMA issued above flies underneath these MFMAs
#if ST_STAGE_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long tB = __builtin_amdgcn_s_memtime();
#endif
            ST_DMA_WAIT(0);                 // this wave's pieces of the next stage have landed ...
#if ST_STAGE_TIMING
            const unsigned long long tC = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();                // ... everyone's have, and everyone is done reading this stage
#if ST_STAGE_TIMING
            {
                const unsigned long long tD = __builtin_amdgcn_s_memtime();
                tm_issue += tA - tS; tm_compute += tB - tA; tm_dma += tC - tB; tm_barrier += tD - tC; tS = tD; ++tm_n;
            }
#endif
            ++it;
        }
    }

#if ST_STAGE_TIMING
    const unsigned long long tLoop = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (EPI == EPI_ACT16 || EPI == EPI_GELU16)
        g2_epilogue_act16<P, BC, BF, WC, WF, EPI == EPI_GELU16>(acc, smem, g, n, t0, BF, cbase, wave, lane);
    else if constexpr (EPI == EPI_QKV) g2_epilogue_qkv<P, BC, BF, WC, WF>(acc, smem, g, n, t0, cbase, wave, lane);
    else g2_epilogue<P, EPI, BC, BF, WC, WF>(acc, (float*)smem, g, kz * g.n_items + n, t0, BF, cbase, wave, lane);
#if ST_STAGE_TIMING
    if (g.dbg && lane == 0 && (wave == 0 || wave == NW - 1) && lin < 64) {
        unsigned long long* d = g.dbg + (size_t)(lin * 2 + (wave ? 1 : 0)) * 8;
        d[0] = tm_issue; d[1] = tm_compute; d[2] = tm_dma; d[3] = tm_barrier; d[4] = tm_n;
        d[5] = tLoop - tStart; d[6] = __builtin_amdgcn_s_memtime() - tLoop; d[7] = 1;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// k = 3 convolution, 128 channels x 126 frames per block, THREE weight buffers: the LDS-DMA of stage g+2 is
// issued at the top of stage g and only a COUNTED s_waitcnt (never vmcnt(0)) + raw s_barrier separate the
// stages, so the DMA of a whole stage stays in flight across each barrier.  The activation tile is exactly
// 128 rows = frames t0-1 .. t0+126 (126 valid output frames + its own halo), which makes the footprint
// 2*16 KB (A) + 3*16 KB (W) = 80 KB: two blocks per CU.  Frame columns 126,127 of the accumulator read two
// rows past the A tile (harmless garbage inside the block's own LDS) and are discarded by the epilogue.
template <class P, int EPI, int BC, int BF, int WC, int WF>
__global__ __launch_bounds__(64 * WC * WF, 2) void conv_gemm3_kernel(const ConvGemmArgs g) {
    using vec8 = typename P::vec8;
    constexpr int BFV = BF - 2, NW = WC * WF, TC = BC / WC, TF = BF / WF, FC = TC / 32, FF = TF / 32;
    constexpr int A_BYTES = BF * 128, W_BYTES = BC * 128;
    constexpr int WPW = (BC / 8) / NW, APW = (BF / 8) / NW;     // 1-KiB DMA pieces per wave per tile
    static_assert(WPW == 4 && APW == 4, "the counted waits below assume 4 + 4 pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                    // 2 buffers
    unsigned char* Ws = smem + 2 * A_BYTES;      // 3 buffers

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
    const int cbase = tc * BC, t0 = tf * BFV;
    if (g.t_lim && t0 >= g.t_lim[n % g.t_lim_mod]) return;      // ragged batch: this tile lies past the item's last needed frame
    const int cin = g.c0 + g.c1;
    const int nch = cin >> 6;
    const int nst = nch * 3;
    const int T = g.T;

    const unsigned char* a0 = (const unsigned char*)g.a0 + (size_t)(n % g.a0_mod) * T * g.c0 * 2;
    const unsigned char* a1 = g.c1 ? (const unsigned char*)g.a1 + (size_t)(n % g.a1_mod) * T * g.c1 * 2 : nullptr;
    const unsigned char* wsrc = (const unsigned char*)g.w;
    const unsigned char* zeros = (const unsigned char*)g.zeros;

    const int prow = lane >> 3;
    auto issueW = [&](int c, int j, int buf) {
#pragma unroll
        for (int k = 0; k < WPW; ++k) {
            const int piece = wave * WPW + k;
            const int row = piece * 8 + prow;
            const int seg = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned char* src = wsrc + ((size_t)((cbase + row) * 3 + j) * cin + (c << 6)) * 2 + seg * 16;
            glds16b(src, Ws + buf * W_BYTES + piece * 1024);
        }
    };
    auto issueA = [&](int c, int buf) {
        const int ch0 = c << 6;
        const unsigned char* srcb; int cs, coff;
        if (ch0 < g.c0) { srcb = a0; cs = g.c0; coff = ch0; }
        else            { srcb = a1; cs = g.c1; coff = ch0 - g.c0; }
#pragma unroll
        for (int k = 0; k < APW; ++k) {
            const int piece = wave * APW + k;
            const int row = piece * 8 + prow;
            const int seg = (lane & 7) ^ ((row >> 1) & 7);
            const int t = t0 + row - 1;
            const unsigned char* src = (t >= 0 && t < T) ? srcb + ((size_t)t * cs + coff) * 2 + seg * 16 : zeros;
            glds16b(src, As + buf * A_BYTES + piece * 1024);
        }
    };

    f32x16_t acc[FC][FF];
    g2_init_acc<EPI, FC, FF>(acc, g, cbase + wc * TC, hi);

    int wrow_off[FC], wswz[FC];
#pragma unroll
    for (int a = 0; a < FC; ++a) {
        const int row = wc * TC + a * 32 + l31;
        wrow_off[a] = row * 128; wswz[a] = (row >> 1) & 7;
    }
    auto compute = [&](int abuf, int wbuf, int j) {
        const unsigned char* Ab = As + abuf * A_BYTES;
        const unsigned char* Wb = Ws + wbuf * W_BYTES;
        int arow_off[FF], aswz[FF];
#pragma unroll
        for (int b = 0; b < FF; ++b) {
            const int row = wf * TF + b * 32 + l31 + j;      // rows 128,129 (last two frame columns) fall into the next buffer
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
#pragma unroll
            for (int a = 0; a < FC; ++a)
#pragma unroll
                for (int b = 0; b < FF; ++b) acc[a][b] = P::mfma(wfr[a], afr[b], acc[a][b]);
        }
    };
    // counted wait: everything except the youngest `keep` LDS-DMA instructions of this wave has landed
    auto wait_keep = [&](int keep) {
        if (keep >= 8)      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (keep >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    issueA(0, 0); issueW(0, 0, 0);
    if (nst > 1) issueW(0, 1, 1);
    wait_keep(0);
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < nch; ++c) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int gi = c * 3 + j;
            int younger = 0;
            if (j == 0 && c + 1 < nch) { issueA(c + 1, (c + 1) & 1); younger += 4; }
            if (gi + 2 < nst) {
                if (j == 0) issueW(c, 2, 2); else issueW(c + 1, j - 1, j - 1);     // stage g+2 lives in buffer (g+2) % 3
                younger += 4;
            }
            compute(c & 1, j, j);
            // stage g+1 needs W(g+1) (issued one stage ago) and, at j == 2, A(c+1) (issued at j == 0): both are
            // older than what this stage issued, except A(c+1) at j == 0 which is not needed before stage (c+1, 0)
            wait_keep(younger);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if constexpr (EPI == EPI_ACT16) g2_epilogue_act16<P, BC, BF, WC, WF>(acc, smem, g, n, t0, BFV, cbase, wave, lane);
    else g2_epilogue<P, EPI, BC, BF, WC, WF>(acc, (float*)smem, g, n, t0, BFV, cbase, wave, lane);
}

// Second half of a split-K launch (cout == 256): one wave per frame row, lane = 4 channels.  Adds the `S` partial
// planes in order (deterministic) and runs the same per-row epilogue the fused kernels run on their accumulators
// (g2_rows: bias / mask / gate / residual, optional FiLM + LayerNorm + modulate of the next op).
template <class P, int EPI>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const ConvGemmArgs g, const float* __restrict__ part, int S) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rows = g.n_items * g.T;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int n = row / g.T, t = row - n * g.T;
    const int ch = lane * 4;
    const G2Consts kc = g2_consts<EPI, true>(g, n, ch);
    const float* src = part + (size_t)row * 256 + ch;
    const size_t plane = (size_t)rows * 256;
    float4 v[1]; v[0] = *(const float4*)src;
    int s = 1;
    for (; s + 4 <= S; s += 4) {       // four planes in flight per round, added in plane order
        float4 p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = *(const float4*)(src + (s + i) * plane);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[0].x += p[i].x; v[0].y += p[i].y; v[0].z += p[i].z; v[0].w += p[i].w; }
    }
    for (; s < S; ++s) {
        const float4 p = *(const float4*)(src + s * plane);
        v[0].x += p.x; v[0].y += p.y; v[0].z += p.z; v[0].w += p.w;
    }
    const float m[1] = {g.mask ? g.mask[(size_t)(n % g.mask_mod) * g.T + t] : 1.0f};
    float4 xin[1]; xin[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI == EPI_RESGATE) xin[0] = *(const float4*)((g.res32 ? g.res32 : g.out32) + (size_t)row * 256 + ch);
    if constexpr (EPI == EPI_F32) {
        if (g.add32) xin[0] = *(const float4*)(g.add32 + ((size_t)(n < g.add_clamp ? n : g.add_clamp) * g.T + t) * 256 + ch);
    }
    const int tt[1] = {t}; const bool ok[1] = {true};
    g2_rows<P, EPI, true, 1>(g, kc, n, tt, ok, ch, v, m, xin);
}

template <class P>
static hipError_t launch_splitk_finish_t(int epi, const ConvGemmArgs& a, const float* part, int S, hipStream_t s) {
    if (a.cout != 256 || S < 1 || !part) return hipErrorInvalidValue;
    const int rows = a.n_items * a.T;
    const dim3 grid((rows + 3) / 4), blk(256);
    if (epi == EPI_F32) hipLaunchKernelGGL((splitk_finish_kernel<P, EPI_F32>), grid, blk, 0, s, a, part, S);
    else if (epi == EPI_RESGATE) hipLaunchKernelGGL((splitk_finish_kernel<P, EPI_RESGATE>), grid, blk, 0, s, a, part, S);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <class P, int EPI, int BC = 128, int BF = 128, int WC = 2, int WF = 2>
static hipError_t launch_g3(const ConvGemmArgs& a, hipStream_t s) {
    constexpr int lds_loop = 2 * BF * 128 + 3 * BC * 128, lds_stage = (WC * WF == 8 ? BF / WF : BF) * (BC + 4) * 4;
    constexpr int lds2 = lds_loop > lds_stage ? lds_loop : lds_stage, lds16 = BF * (BC * 2 + 16);
    constexpr int lds = lds2 > lds16 ? lds2 : lds16;
    // the >64 KB dynamic-LDS opt-in is per device: remember it per device id (engines may live on several GPUs)
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    bool& attr_done = attr_done_dev[dev_];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_gemm3_kernel<P, EPI, BC, BF, WC, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (!a.zeros || (a.cout % BC) != 0 || ((a.c0 | a.c1) & 63) != 0 || a.c2 != 0 || a.ksplit > 1) return hipErrorInvalidValue;
    ConvGemmArgs b = a;
    b.tiles_f = (a.T + BF - 3) / (BF - 2);
    b.tiles_c = a.cout / BC;
    const int total = b.n_items * b.tiles_f * b.tiles_c;
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL((conv_gemm3_kernel<P, EPI, BC, BF, WC, WF>), dim3(grid), dim3(64 * WC * WF), lds, s, b);
    return hipGetLastError();
}

template <class P, int TAPS, int EPI, int BC, int BF, int WC, int WF, int WBUF = 2>
static hipError_t launch_g2(const ConvGemmArgs& a, hipStream_t s) {
    using K = G2Cfg<BC, BF, WC, WF, TAPS, WBUF>;
    constexpr int qkv_lds = (EPI == EPI_QKV) ? g2_qkv_lds_bytes<BC, BF>() : 0;
    constexpr int LDS = K::LDS_BYTES > qkv_lds ? K::LDS_BYTES : qkv_lds;
    // the >64 KB dynamic-LDS opt-in is per device: remember it per device id (engines may live on several GPUs)
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    bool& attr_done = attr_done_dev[dev_];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_gemm2_kernel<P, TAPS, EPI, BC, BF, WC, WF, WBUF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (!a.zeros || (a.cout % BC) != 0 || ((a.c0 | a.c1 | a.c2) & 63) != 0 || (a.c2 && a.c2 > a.c0)) return hipErrorInvalidValue;
    ConvGemmArgs b = a;
    b.tiles_f = (a.T + BF - 1) / BF;
    b.tiles_c = a.cout / BC;
    if (a.ksplit > 1) {      // raw partial sums only: no bias / mask / residual / 16-bit outputs, every split owns >= 1 chunk
        if (EPI != EPI_F32 || a.bias || a.add32 || a.out16 || a.ln_h16 || (a.flags & GF_MASK) || !a.out32 ||
            a.ksplit > (a.c0 + a.c1 + a.c2) / 64 || a.w_item_stride)
            return hipErrorInvalidValue;
    }
    const int total = b.n_items * b.tiles_f * b.tiles_c * (a.ksplit > 1 ? a.ksplit : 1);
    const int grid = ((total + 7) / 8) * 8;
    if (EPI == EPI_QKV && (a.cout != 3 * BC || a.n_heads * 64 != BC || !a.q || !a.k || !a.vt)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_gemm2_kernel<P, TAPS, EPI, BC, BF, WC, WF, WBUF>), dim3(grid), dim3(K::NT), LDS, s, b);
    return hipGetLastError();
}

}  // namespace st
