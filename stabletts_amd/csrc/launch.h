// Host-visible launcher interface between engine.cpp and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace st {

enum { DT_BF16 = 0, DT_F16 = 1 };

// ---------------------------------------------------------------- implicit-GEMM convolution
// out[ch][frame] = bias[ch] + sum_{tap j, ci} W[ch][j][ci] * act[frame + j - TAPS/2][ci]
// Activations are TIME-MAJOR 16-bit tensors [item][T][C]; the K dimension may span two source
// tensors (channel concat without materialising torch.cat: estimator.py:120,131).
enum { EPI_ACT16 = 0, EPI_F32 = 1, EPI_RESGATE = 2, EPI_QKV = 3,
       EPI_GELU16 = 4,     // EPI_ACT16 with exact-erf GELU in place of SiLU (Vocos pwconv1, module.py:38-39)
       EPI_SILU = 5 };     // training FFN (k = 3, phased kernel only): EPI_F32 + the SiLU / dropout step fused, see act16 / dact16
enum { GF_SILU = 1, GF_MASK = 2,
       GF_K2_V_ONLY = 4 };     // EPI_QKV with a second source: its weights are zero in the q and k rows (training: v = W_v h_hi + W_v h_lo) -- q / k blocks stop at c0

struct ConvGemmArgs {
    const void* a0; const void* a1;   // activation sources [items][T][c0], [items][T][c1]
    int c0, c1;                       // channels per source (multiples of 64; c1 may be 0)
    int c2;                           // further K channels that re-read a0 from its channel 0 (split-precision operands:
                                      // K = [a0 | a1 | a0] against packed weights [W_hi | W_hi | W_lo]); gen-2 kernels only
    int a0_mod, a1_mod;               // activation item of output item n is n % mod
    const void* w;                    // packed weights [cout][TAPS][c0 + c1 + c2], 16-bit
    long long w_item_stride;          // bytes added to w per output item (0: shared weights; weight-gradient GEMMs use it)
    const float* bias;                // [cout] or nullptr
    int cout, T, n_items;
    int tiles_f, tiles_c;             // frame tiles per item, channel tiles
    const float* mask; int mask_mod;  // [mask_mod][T] float 0/1, item n -> n % mask_mod
    int flags;                        // GF_*
    void* out16; float* out32;        // [items][T][cout]
    int out32_readonly;               // EPI_RESGATE: 1 = out32 is only READ (the residual); the updated fp32 rows are not stored
    void* out16_lo;                   // optional (with out16, fp32-staged epilogues): 16-bit residual x - float(out16)
    const float* add32; int add_clamp;  // EPI_F32: + add32[min(n, add_clamp)][t][ch]
    const float* gate; int gate_stride; // EPI_RESGATE: out32 += gate[n*gate_stride + ch]*((acc+b)*mask)
    const float* res32;               // EPI_RESGATE: the residual is read from res32 instead of out32 (nullptr: in place) -- training keeps x1 / x2 apart
    float* branch32;                  // EPI_RESGATE: also stores the branch output (acc+b)*mask in fp32 (training: the gate's gradient needs it)
    // EPI_QKV (cout = 3*C, head_dim 64): q,k -> [item][H][T][64], vT -> [item][H][64][Tp]
    void* q; void* k; void* vt;
    void* q_lo; void* k_lo;           // split-precision attention operands (nullptr: off): the rounding residuals of q and k, same layout
    void* vt_lo;                      // training (nullptr: off): the rounding residual of v, layout of vt -- the attention output is then P (v_hi + v_lo)
    const float* rope_cos; const float* rope_sin;   // [T][16]
    int Tp; float qscale; int n_heads;
    const void* zeros;                // >= 16 bytes of zeros in global memory (halo source of the LDS-DMA path)
    const void* w_frag;               // qkv_ws.hip: the q/k/v weight in fragment order (launch_pack_qkv_frag)
    void* sink;                       // >= 64 KiB of scratch that rows outside the tensor are stored to (qkv_ws.hip: every wave issues a FIXED number of stores)
    // fused prologue of the NEXT op (row-complete tiles, cout == 256): FiLM -> *mask -> LayerNorm -> modulate.
    // When ln_h16 != nullptr, out32 receives the post-FiLM residual stream and ln_h16 the 16-bit operand.
    void* ln_h16;
    const float* ln_film; int ln_film_stride; int ln_film_mod;     // gamma = film[(n%mod)*stride + ch], beta = +256
    const float* ln_ada; int ln_ada_stride; int ln_shift_off; int ln_scale_off;
    int ln_mask_out;
    // Ragged batches: frame tiles of item n that start at or beyond t_lim[n % t_lim_mod] are not computed at all
    // (nullptr: every tile).  t_lim = last valid frame + 1 + kFrameHalo (mask_prep), so every frame a VALID output
    // frame depends on -- including the reference's pad leak through the unmasked tensors, SURVEY A.5 -- is computed.
    const int* t_lim; int t_lim_mod;
    // EPI_SILU (training FFN, diffusion_transformer.py:25-30).  Forward (act16 != nullptr): out16 = the pre-activation, act16 =
    // silu(float(out16)) * dropout factor * mask -- what silu_drop_kernel computes from out16.  Backward (dact16 != nullptr):
    // out16 = acc * mask * dropout factor * silu'(float(dact16[row][ch])) -- what silu_bwd_kernel computes from the fp32 result.
    // Same arithmetic order as the two stand-alone kernels: results are bit-identical.  Dropout = the FFN element-pair hash.
    void* act16; const void* dact16;
    unsigned long long drop_seed; unsigned drop_thresh16; float drop_scale;
    int ksplit;                       // > 1: split-K launch (EPI_F32 only): out32 = partial planes [ksplit][items][T][cout], raw sums
    // fused FFN (ffn_fused.h): w = the fused weight stream (launch_pack_ffn_stream), bias1 = conv_1's bias [cmid], cmid = the
    // intermediate width (filter_channels); every other field describes conv_2's EPI_RESGATE epilogue, a0 = the FFN input
    const float* bias1; int cmid;
    unsigned long long* dbg;          // diagnostics (ST_STAGE_TIMING builds of tools/gemm2_bench only), else nullptr
};

// tile configurations (conv_gemm2_impl.h): T128 = 128x128 tile, RC = row-complete 256x128 tile (cout % 256 == 0,
// may carry the fused FiLM + LayerNorm + modulate of the next op through the ln_* fields; also the QKV epilogue)
enum { G2_T128 = 0, G2_RC = 1, G2_K3PIPE = 2,   // K3PIPE: k=3 only, three weight buffers, counted vmcnt
       G2_BIG = 3,                               // 256 x 256 tile, 8 waves of 128 x 64 (cout % 256 == 0)
       G2_T64 = 4, G2_RC64 = 5,                  // 64-frame versions of T128 / RC (QKV only) for small, latency-bound grids
       G2_PHASED = 6,                            // BIG tile, phased K loop (conv_gemm_phased.h)
       G2_RC1 = 7 };                             // RC tile with ONE weight buffer: 74 KB of LDS -> two resident blocks (fused q/k/v projection)
hipError_t launch_conv_gemm2_bf16(int cfg, int taps, int epi, const ConvGemmArgs& a, hipStream_t s);
hipError_t launch_conv_gemm2_f16(int cfg, int taps, int epi, const ConvGemmArgs& a, hipStream_t s);
// epilogue `epi` (EPI_F32 / EPI_RESGATE, cout == 256) of the sum of the S partial planes a split-K launch left in `part`
hipError_t launch_splitk_finish_bf16(int epi, const ConvGemmArgs& a, const float* part, int S, hipStream_t s);
hipError_t launch_splitk_finish_f16(int epi, const ConvGemmArgs& a, const float* part, int S, hipStream_t s);
// Fused FFN (diffusion_transformer.py:20-30 as ONE kernel, the 1024-wide intermediate never leaves the CU): conv_1 + SiLU + mask
// -> 16-bit u chunk in LDS -> conv_2 accumulating over the chunks -> conv_2's EPI_RESGATE(+LN) epilogue.  a.w = fused weight
// stream, a.bias1 / a.cmid = conv_1 bias / width, everything else as for conv_2 (a0 = the FFN input h2, c0 = cout = 256).
constexpr int kFfnFusedFrames = 126;      // output frames per block (128 u rows = the tile + conv_2's halo)
hipError_t launch_ffn_fused_bf16(const ConvGemmArgs& a, hipStream_t s);
hipError_t launch_ffn_fused_f16(const ConvGemmArgs& a, hipStream_t s);
// Weight stream of the fused FFN kernel, in the order the kernel consumes it: for each 256-channel chunk c of the intermediate
// width, 24 conv_1 slabs (cin chunk, tap, k-step pair) then 24 conv_2 slabs (u sub-chunk, tap, k-step pair); a slab = 16 MFMA
// A-fragments of 1 KiB stored lane-linear (fragment (ksl, a8): rows a8*32.., k-step 2*kp+ksl).  stage 0: src = conv_1 weight
// (F, 256, 3); stage 1: src = conv_2 weight (256, F, 3).  (common.h: ffn_stream_index)
hipError_t launch_pack_ffn_stream(int dtype, const float* src, int stage, int F, void* dst, hipStream_t s);
hipError_t launch_pack_ffn_wino(const float* src, int stage, int F, void* dst, hipStream_t s);      // ffn_wino.h's stream (f16)
hipError_t launch_ffn_wino_f16(const ConvGemmArgs& a, hipStream_t s);                                   // Winograd F(2,3) fused FFN, f16 operands only
// Fused q / k / v projection + RoPE of big grids as a weight-stationary persistent kernel (qkv_ws.hip): same arguments and results
// (bit for bit) as launch_conv_gemm2_*(G2_RC*, 1, EPI_QKV, ...); needs hidden = 256, 4 heads, a.sink.
hipError_t launch_qkv_ws(int dtype, const ConvGemmArgs& a, hipStream_t s);
// The attention out-projection + gate + residual + LayerNorm_2 + modulate of big grids as a weight-stationary persistent kernel
// (oproj_ws.hip): same arguments and results (bit for bit) as launch_conv_gemm2_*(G2_BIG, 1, EPI_RESGATE, ...) with ln_h16 set, no
// FiLM, the residual updated in place; needs hidden = 256, a.w_frag (plane 0 of a launch_pack_qkv_frag copy), a.sink.
hipError_t launch_oproj_ws(int dtype, const ConvGemmArgs& a, hipStream_t s);
// plane (0 q, 1 k, 2 v) of the fragment-ordered weight copy that kernel reads: src = fp32 conv weight (256, 256, 1) of the plane,
// dst = the whole 3 x 256 x 256 16-bit buffer (common.h: qkv_frag_index).  PackJob kind 4 (row_off = 256 * plane) is the same map.
hipError_t launch_pack_qkv_frag(int dtype, const float* src, int plane, void* dst, hipStream_t s);
constexpr int kGemmFramesPerTile = 128;
constexpr int kGemmChannelsPerTile = 128;

// ---------------------------------------------------------------- counter-based dropout (training; shared by forward and backward)
// One 32-bit hash decides TWO neighbouring elements (its low / high 16 bits against thresh16 = p * 2^16); kept values are scaled
// by 1 / (1 - p) like nn.Dropout and SDPA's dropout_p.  thresh16 == 0: dropout off.
//   FFN sites (element index i):         h = mix32(((u32)seed ^ (i >> 1) * 0x9E3779B1) + (u32)(seed >> 32)), half = i & 1
//   attention sites (row, key):          h = pair(rowh[row], colh[key >> 1]), half = key & 1, where the two tables
//                                        rowh[row] = mix32((u32)seed ^ row * 0x9E3779B1), colh[j] = mix32((u32)(seed >> 32) ^ j * 0x85ebca77)
//                                        are filled once per attention call (launch_drop_tables): whichever of (row, key) a
//                                        kernel has on its lanes, the per-element work is one 2-multiply mix of rowh ^ colh
struct DropCfg { unsigned long long seed; unsigned thresh16; float scale; const unsigned* rowh; const unsigned* colh; };
hipError_t launch_drop_tables(const DropCfg& d, int n_rows, int n_colpairs, unsigned* rowh, unsigned* colh, hipStream_t s);

// ---------------------------------------------------------------- attention
struct AttnArgs {
    const void* q; const void* k; const void* vt; void* out;   // out: [item][T][H*64] 16-bit
    const void* q_lo; const void* k_lo;    // inference: split-precision scores s = q_hi k_hi + q_lo k_hi + q_hi k_lo (nullptr: s = q k); 16-bit, layout of q / k
    const void* vt_lo;                     // training: the rounding residuals of v (layout of vt): O = P v_hi + P v_lo (nullptr: O = P v)
    unsigned* lse_max;                     // inference: a group of kLseCells cells 64 B apart receiving max over rows of the log2-sum-exp (order-preserving int bits; nullptr: off)
    const float* kbias; int mask_mod;      // additive key bias [mask_mod][Tp]: 0 valid, -1e30 masked / >= T
    const int* kv_end; const int* n_full;  // per mask row: last valid key + 1, leading valid prefix length
    int T, Tp, H, n_items;
    const void* zeros;                     // >= 16 zero bytes in global memory (out-of-range K rows)
    float* lse;                            // training: [item][H][T] log2-sum-exp of the scores (nullptr: inference kernel)
    DropCfg drop;                          // training: dropout on the attention probabilities (diffusion_transformer.py:77)
    const int* t_lim;                      // query tiles of mask row mb that start at or beyond t_lim[mb] are skipped (nullptr: none)
    int small_max_blocks;                  // inference: launches of <= this many 256-query blocks use the key-split small-grid kernel (0: never)
};
constexpr int kLseCells = 16;
hipError_t launch_attention(int dtype, const AttnArgs& a, hipStream_t s);

// ---------------------------------------------------------------- FiLM + LayerNorm + adaLN modulate
struct FilmLnArgs {
    float* X;                   // residual stream [rows][256] fp32 (updated in place when film != nullptr)
    void* h16;                  // modulated LayerNorm output [rows][256] 16-bit
    const float* film; int film_stride; int film_mod;   // gamma = film[(n%mod)*stride + ch], beta = +256; or nullptr
    const float* ada; int ada_stride;                   // shift = ada[n*stride + shift_off + ch], scale = + scale_off
    int shift_off, scale_off;
    const float* mask; int mask_mod;
    int mask_out;               // multiply h by the mask (FFN input, diffusion_transformer.py:26)
    int T, rows;
};
hipError_t launch_film_ln(int dtype, const FilmLnArgs& a, hipStream_t s);

// ---------------------------------------------------------------- small fp32 helpers
hipError_t launch_time_embed(const float* t, int n_t, int dim, float* emb, hipStream_t s);
// out[n][o] = act_out(bias[o] + sum_i W[o][i] * act_in(in[n][i]))
hipError_t launch_silu_rows(const float* in, int64_t n, float* out, hipStream_t s);      // out = SiLU(in), fp32
struct LinearJobs { const float* in[8]; const float* W[8]; const float* bias[8]; float* out[8]; int n; };
hipError_t launch_linear_multi(const LinearJobs& J, int n, int k, int o, int silu_in, int silu_out, hipStream_t s);
hipError_t launch_linear(const float* in, int n, int k, const float* W, const float* bias, int o,
                         float* out, int silu_in, int silu_out, hipStream_t s);
// t_lim (optional, B + 1 ints): t_lim[b] = min(T, kv_end[b] + kFrameHalo), t_lim[B] = max over the rows
constexpr int kFrameHalo = 4;   // frames past the last valid one that valid outputs depend on: in_proj's long skip reads cond one
                                // frame out (k = 3), cond = three k = 3 convs of the unmasked mu (estimator.py:83-89,118,131)
hipError_t launch_mask_prep(const float* mask, int B, int T, int Tp, int* n_full, int* kv_end, float* kbias, int* t_lim, hipStream_t s);

// (B, C, T) fp32 -> time-major (B, T, Cp): fp32 and/or 16-bit (+ optional 16-bit rounding residual out16lo),
// channels >= C zero-filled
hipError_t launch_to_time_major(int dtype, const float* in, int B, int C, int T, int Cp,
                                float* out32, void* out16, void* out16lo, hipStream_t s);
// time-major (B, T, Cp) fp32 -> (B, C, T) fp32
hipError_t launch_from_time_major(const float* in, int B, int C, int T, int Cp, float* out, hipStream_t s, int* nonfinite = nullptr);
// dst16[t][c] = vec[c] for all t (uncond prenet input: fake_content broadcast, flow_matching.py:60)
hipError_t launch_embed_tokens(const long long* tokens, const long long* lengths, const float* emb, int n_vocab,
                               int C, float scale, int B, int T, float* X, float* mask_out, hipStream_t s);
hipError_t launch_set_values(float* dst, const float* host_vals, int n, hipStream_t s);
hipError_t launch_delay(int us, hipStream_t s);      // one sleeping wave holds stream s for ~us microseconds (phase offset between solve parts)
hipError_t launch_cvec_prep(const float* c, const float* fake_or_null, int B, int G, float* dst, hipStream_t s);
hipError_t launch_fill_rows16(int dtype, const float* vec, int C, int Cp, int T, void* out16, hipStream_t s);

// v: [N2][T][Cp] estimator outputs. If use_cfg: vv = v[B+b] + s*(v[b] - v[B+b]) else vv = v[b].
// kout (optional) = vv ; if xio: xio += dt*vv and (x16, x16lo) = split-precision operand pair of xio (Euler step fused).
hipError_t launch_cfg_combine(int dtype, const float* v, int B, int64_t per_item, int use_cfg, float s,
                              float* kout, float* xio, void* x16, void* x16lo, float dt, hipStream_t stream);
// y = x + sum_i coef[i]*k[i] (i < nk <= 7); writes y32 and/or the operand pair (y16, y16lo)
hipError_t launch_lincomb(int dtype, const float* x, const float* const* k, const float* coef, int nk,
                          int64_t n, float* y32, void* y16, void* y16lo, hipStream_t s);

// weight packing: src fp32 (cout, cin_total, K) -> dst16 [cout_p][K][cin_p], columns [col_off, col_off + slice_w):
// source channels [ci_off, ci_off + ci_cnt) then zeros.  dst row offset `row_off` (QKV concat).  lo != 0 packs the
// rounding residual W - float(to16(W)) instead of W (split-precision weights).
hipError_t launch_pack_weight(int dtype, const float* src, int cout, int cin_total, int K, int ci_off,
                              int ci_cnt, void* dst, int row_off, int cin_p, int col_off, int slice_w, int lo,
                              hipStream_t s);
hipError_t launch_cvt16_to_f32(int dtype, const void* src, float* dst, int64_t n, hipStream_t s);
// One launch for a whole list of packing jobs (the ~100 launch_pack_weight / launch_pack_weight_t calls and ~60 bias copies of a
// re-pack after an optimizer step are each a few microseconds of launch latency: 0.75 ms per training step as separate launches).
// kind 0: launch_pack_weight's mapping, kind 1: launch_pack_weight_t's (train_launch.h), kind 2: fp32 copy of `cout` elements,
// kind 3: launch_pack_ffn_stream's (lo = stage, cout = F), kind 4: launch_pack_qkv_frag's (row_off = 256 * plane),
// kind 5: launch_pack_ffn_wino's (lo = stage, cout = F).
// blk0 = first 256-thread block of the job in the merged grid (jobs sorted by blk0).
struct PackJob { const float* src; void* dst; int kind, cout, cin_total, K, ci_off, ci_cnt, row_off, cin_p, col_off, slice_w, lo; unsigned blk0; };
hipError_t launch_pack_jobs(int dtype, const PackJob* jobs_dev, int njobs, unsigned nblocks, hipStream_t s);

// ---------------------------------------------------------------- compute_loss's own arithmetic (flow_matching.py:86-100)
constexpr int kCfmLossBlocks = 1024;       // scratch of launch_cfm_loss: 2 * kCfmLossBlocks + 2 floats
hipError_t launch_cfm_loss_prep(const float* x1, const float* z, const float* t_rand, float sigma_min, int B, int M, int T,
                                float* t_out, float* y, float* u, hipStream_t s);
hipError_t launch_cfm_loss(const float* pred, const float* u, const float* mask, int B, int M, int T, float* scratch, float* loss, hipStream_t s);
hipError_t launch_cfm_loss_bwd(const float* pred, const float* u, const float* scratch, const float* grad_loss, int B, int M, int T, float* gpred, hipStream_t s);

// ---------------------------------------------------------------- duration -> alignment -> mu_y (align_kernels.hip; models/model.py:17-27,82-96)
hipError_t launch_durations(const float* logw, const float* x_mask, float length_scale, int B, int Tx, float* w_ceil,
                            float* cum, long long* y_len, hipStream_t s);
hipError_t launch_cumsum_rows(const float* dur, int B, int Tx, float* cum, hipStream_t s);
hipError_t launch_path(const float* cum, const float* mask, int B, int Tx, int Ty, float* path, hipStream_t s);
hipError_t launch_align(const float* cum, const float* x_mask, const long long* y_len, const float* mu_x, int B, int M, int Tx,
                        int Ty, float* attn, float* mu_y, float* y_mask, hipStream_t s);

// ---------------------------------------------------------------- adaptive dopri5 support (adaptive_ode.hip)
hipError_t launch_set_scalar(float* dst, float v, hipStream_t s);
// Deterministic sums of squares over n elements (two-stage reduction, fixed order), result in out[0..1]:
//   mode 0: out[0] = sum (y/scale)^2, out[1] = sum (f/scale)^2            scale = atol + rtol*|y|      (a=y, b=f)
//   mode 1: out[0] = sum ((b - a)/scale)^2                                scale = atol + rtol*|y|      (a=f0, b=f1)
//   mode 2: out[0] = sum (err/tol)^2, err = sum_j coef[j]*k[j], tol = atol + rtol*max(|y|,|y1|)      (a = y1)
//   mode 3: out[0] = number of elements with |a - b| / (atol + rtol*max(|a|,|b|)) not < 1   (implicit Adams corrector; y = any valid pointer)
struct OdeNormArgs { const float* y; const float* a; const float* b; const float* k[7]; float coef[7]; int nk;
                     float rtol, atol; int64_t n; int mode; float* partial; float* out; };
hipError_t launch_ode_norm(const OdeNormArgs& a, hipStream_t s);
constexpr int kOdeNormBlocks = 1024;
// dense output of dopri5 at x = (t - t0)/(t1 - t0): out = a x^4 + b x^3 + c x^2 + d x + y0 (torchdiffeq _interp_fit)
hipError_t launch_dopri5_interp(const float* y0, const float* y1, const float* const* k, const float* cmid_dt, float dt,
                                float x, int64_t n, float* out, hipStream_t s);

}  // namespace st
