// Launcher interface of the training-only kernels (train_kernels.hip, attention_bwd.hip): the forward pieces
// that keep activations, the backward elementwise / reduction kernels, the operand transposes that turn the
// weight gradient into a call of the forward implicit-GEMM kernel, and flash-attention backward.
// Reference: autograd through models/estimator.py:103-138 and models/diffusion_transformer.py:25-121 as driven by
// CFMDecoder.compute_loss (models/flow_matching.py:69-100) under train.py:78-81.
#pragma once
#include "launch.h"

namespace st {

// DropCfg of one dropout site (salt = site index: FFN / attention of block i) for probability p
DropCfg make_drop(float p, unsigned long long seed, int salt);

// ---------------------------------------------------------------- forward pieces (one frame per wave, lane = 4 channels, C = 256)
// x = FiLM:  (gamma * xin + beta) * mask      (estimator.py:16,31-33)
//     RES :  xin + gate * branch              (diffusion_transformer.py:111-112; branch already masked)
//     else:  xin
// writes xout (fp32), optional 16-bit copies x16 (+ rounding residual x16lo), and, when h16 != nullptr,
// h16 = (LayerNorm(x) * (1 + scale) + shift) [* mask if mask_out]
struct TrainLnArgs {
    const float* xin; float* xout; void* x16; void* x16lo; void* h16;
    void* h16lo;                // optional: the rounding residuals of h16 (ST_TRAIN_VLO=2: the v projection reads h as a hi + lo pair)
    const float* film; int film_stride; int film_mod;
    const float* gate; int gate_stride; const float* branch;
    const float* ada; int ada_stride; int shift_off; int scale_off;
    const float* mask; int mask_mod; int mask_out;
    int T, rows;
};
hipError_t launch_train_ln(int dtype, const TrainLnArgs& a, hipStream_t s);

// u16 = dropout(SiLU(a16)) * mask   over [rows][F]   (FFN: diffusion_transformer.py:26-28; prenet: mask == nullptr)
hipError_t launch_silu_drop(int dtype, const void* a16, void* u16, const float* mask, int mask_mod, int T, int F,
                            int64_t rows, DropCfg drop, hipStream_t s);

// ---------------------------------------------------------------- backward elementwise + per-item channel reductions
// Every kernel below walks blocks of kRedRows frames of one item; per-(item, channel) sums are accumulated in
// registers (a lane owns the same 4 channels for every frame), combined per block and written to
// part[item][chunk][k][256]; launch_reduce_parts adds the chunks in order (deterministic, no atomics).
constexpr int kRedRows = 32;
inline int red_chunks(int T) { return (T + kRedRows - 1) / kRedRows; }
// out[n][k][256] (+)= sum_chunk part[n][chunk][k][256]
hipError_t launch_reduce_parts(const float* part, int n_items, int chunks, int K, float* out, int out_stride,
                               const int* out_off, int accumulate, const float* unscale, hipStream_t s);

// Gradient scaling (see train_kernels.hip): sc[0] = power-of-two scale that puts max |g| at ~2^8, sc[1] = 1 / sc[0].
// Kernels that take `scale` multiply by sc[0]; kernels that take `unscale` multiply by sc[1] (nullptr: 1).
hipError_t launch_grad_scale(const float* g, int64_t n, unsigned* bits, float* sc, hipStream_t s);
hipError_t launch_unscale_inplace(float* a, int64_t n, const float* sc, hipStream_t s);
// Block-boundary re-centring of the pass-wide scale: if max |g| (g = the running fp32 gradient, in scaled units) has left
// [2^4, 2^12), sc[2] = the power of two that brings it back to ~2^8 (else 1), and sc[0] *= sc[2], sc[1] = 1 / sc[0].
// launch_scale_by then multiplies a tensor that lives in scaled units by sc[2] (returns at once when it is 1).
hipError_t launch_grad_rescale(const float* g, int64_t n, unsigned* bits, float* sc, hipStream_t s);
hipError_t launch_scale_by(float* a, int64_t n, const float* sc, hipStream_t s);
// The same re-centring as ONE launch (+ one when no producer published the maximum): `cells` is a group of kMaxCells maxima 64
// bytes apart (zeroed by the caller; producers: ln_bwd / add_rescaled `amax`, or the absmax pass run here when !have_max);
// a *= f and sc_next = {sc[0] f, 1 / (sc[0] f), f}.  sc is only read: kernels enqueued later are handed sc_next.
constexpr int kMaxCells = 16;
constexpr int kMaxCellWords = kMaxCells * 16;
hipError_t launch_recentre(float* a, int64_t n, unsigned* cells, bool have_max, const float* sc, float* sc_next, hipStream_t s);
// every per-(item, channel) sum of one backward block in one launch (replaces one launch_reduce_parts per row kernel)
struct RedSite { const float* part; int K; float* out; int out_stride; int off[2]; const float* unscale; };
struct RedSites { RedSite s[6]; int n; };
hipError_t launch_reduce_sites(const RedSites& S, int n_items, int chunks, hipStream_t s);
// dropout hash tables of every attention site of a forward in one launch
struct DropSeeds { unsigned long long seed[16]; };
hipError_t launch_drop_tables_multi(const DropSeeds& sd, int n_sites, int n_rows, int n_colpairs, unsigned* rowh, unsigned* colh,
                                    size_t row_stride, size_t col_stride, hipStream_t s);

// x_out = x_in + gate * branch:   d branch16 = dX * gate * mask ; part[.][0] = sum_t dX * branch
hipError_t launch_gate_bwd(int dtype, const float* dX, const float* branch, const float* gate, int gate_stride,
                           const float* mask, int mask_mod, int T, int n_items, void* dB16, float* part, hipStream_t s);
// h = (LN(x) (1 + sc) + sh) [* mask]:  dX += LN'(dH ...) ; part[.][0] = d scale, part[.][1] = d shift
// dh_scale (optional): dH is multiplied by dh_scale[1] on load (a GEMM result computed from locally re-scaled operands)
hipError_t launch_ln_bwd(const float* x, const float* dH, const float* ada, int ada_stride, int scale_off,
                         const float* mask, int mask_mod, int mask_out, int T, int n_items, float* dX, float* part,
                         const float* dh_scale, unsigned* amax, hipStream_t s);
// x = (gamma * xpre + beta) * mask:  part[.][0] = d gamma, part[.][1] = d beta ; dX = dX * mask * gamma (in place) (+ 16-bit copy)
hipError_t launch_film_bwd(int dtype, const float* xpre, const float* film, int film_stride, int film_mod,
                           const float* mask, int mask_mod, int T, int n_items, float* dX, void* dX16, float* part,
                           hipStream_t s);
// dA16 = dU * mask * dropout * SiLU'(a16)   over [rows][F]
hipError_t launch_silu_bwd(int dtype, const float* dU, const void* a16, const float* mask, int mask_mod, int T, int F,
                           int64_t rows, DropCfg drop, void* dA16, hipStream_t s);
// y16 = to16(x * mask?)  /  y = a + b  /  fp32 row-slices
hipError_t launch_cast16(int dtype, const float* x, const float* mask, int mask_mod, int T, int C, int64_t rows,
                         const float* scale, void* y16, hipStream_t s);
hipError_t launch_add_inplace(float* a, const float* b, int64_t n, hipStream_t s);
// a += b * sc[0] / sc_b[0]: b was written when the pass-wide scale was sc_b[0], a lives at the current scale sc[0] (both powers of two)
// amax != null: also publishes max |a| after the add into the cell group (launch_recentre's input)
hipError_t launch_add_rescaled(float* a, const float* b, int64_t n, const float* sc, const float* sc_b, unsigned* amax, hipStream_t s);
hipError_t launch_copy_scalars(float* dst, const float* src, int n, hipStream_t s);

// ---------------------------------------------------------------- weight gradient as a forward GEMM
// dW[co][j][ci] = sum_{n,t} dY[n][t][co] * X[n][t + j - taps/2][ci].  With K-contiguous transposed copies
//   XT [S][taps*Cin][Rs]  (row j*Cin + ci, column r_local; zero where the shifted frame leaves the item or r >= R)
//   dYT[S][Cout][Rs]
// split over S chunks of the N*T rows, it is exactly the forward kernel's contraction (taps = 1, "weights" = dYT
// with a per-item stride, "activation frames" = the taps*Cin rows of XT): partial[S][taps*Cin][Cout] fp32.
hipError_t launch_wgrad_xt(int dtype, const void* x0, int c0, const void* x1, int c1, int n_items, int T, int taps,
                           int S, int Rs, void* xt, hipStream_t s);
// also part_b[rowblock][Cout] = column sums of the 64-row block (bias gradient partials)
hipError_t launch_wgrad_dyt(int dtype, const void* dy, int cout, int64_t R, int S, int Rs, void* dyt, float* part_b,
                            hipStream_t s);
// The same weight gradient WITHOUT the transposed copies (wgrad_tn.hip): a "TN" GEMM that stages dY and X in LDS as they lie
// in memory ([frame][channel]) and reads the k-strided MFMA fragments with ds_read_b64_tr_b16.  Splits K over ranges of 32-frame chunks (cps chunks
// per block; a chunk lies inside one item): partial[S = ceil(n_items * ceil(T / 32) / cps)][taps*Cin][cout].  Needs cout % 256 == 0, c0 % 64 == c1 % 64 == 0.
// part_b != null: the blocks of the first N tile also write part_b[S][cout] = column sums of dY over their K range (bias-gradient
// partials as a by-product of the A fragments they already hold; replaces the colsum_rows pass over dY).
hipError_t launch_wgrad_tn(int dtype, const void* dy, int cout, const void* x0, int c0, const void* x1, int c1, int taps,
                           int n_items, int T, int cps, const void* zeros, float* partial, float* part_b, hipStream_t s);
// ONE launch for every output of a weight-gradient GEMM: up to three (co_start, co_cnt) row blocks of the fused q/k/v projection, each
// with its own un-scaling pair; dW as launch_wgrad_reduce, db[co - co_start] = sum_s part_b[s][co] (part_b from launch_wgrad_tn).
struct WgradRed { float* dW; float* db; const float* unscale; int cin_total, ci_off, ci_cnt, co_start, co_cnt; };
hipError_t launch_wgrad_reduce_multi(const float* partial, const float* part_b, int S, int cin, int cout, int taps, const WgradRed* outs,
                                     int n_outs, hipStream_t s);
// part_b[rowblock][cout] = column sums of every 64-row block of dY [R][cout] (bias-gradient partials for launch_bias_reduce)
hipError_t launch_colsum_rows(int dtype, const void* dy, int cout, int64_t R, float* part_b, hipStream_t s);
// dW (reference layout (co_cnt, Cin_total, taps), fp32) [co - co_start][ci_off + ci][j] = sum_s partial[s][j*Cin + ci][co]
// for co in [co_start, co_start + co_cnt) (a row block of a fused projection), ci < ci_cnt
hipError_t launch_wgrad_reduce(const float* partial, int S, int cin, int cout, int taps, float* dW, int cin_total,
                               int ci_off, int ci_cnt, int co_start, int co_cnt, const float* unscale, hipStream_t s);
// db[co - co_start] = sum_rb part_b[rb][co]
hipError_t launch_bias_reduce(const float* part_b, int rowblocks, int cout, float* db, int co_start, int co_cnt,
                              const float* unscale, hipStream_t s);

// dgrad weights: dst16 [Cin_p][taps][ld] columns [col_off, col_off + cout): Wd[ci][j][co] = W[co][ci_off + ci][taps - 1 - j]
hipError_t launch_pack_weight_t(int dtype, const float* src, int cout, int cin_total, int taps, int ci_off, int ci_cnt,
                                void* dst, int cin_p, int ld, int col_off, hipStream_t s);

// ---------------------------------------------------------------- small fp32 linears (adaLN, FiLM, time MLP): backward
// out = W act(in) + b:  dW[o][k] (+)= sum_n dout[n][o] act(in[n][k]);  db[o] (+)= sum_n dout[n][o]
// several layers of one shape per launch (the per-item linears of the blocks of one backward part)
struct LinBwdJobs { const float* in[8]; const float* dout[8]; const float* W[8]; float* dW[8]; float* db[8]; int n; };
hipError_t launch_linear_bwd_w_multi(const LinBwdJobs& J, int n, int k, int o, int silu_in, hipStream_t s);
hipError_t launch_linear_bwd_in_multi(const LinBwdJobs& J, const float* in, int n, int k, int o, int silu_in, float* din, int accumulate, hipStream_t s);
hipError_t launch_linear_bwd_w(const float* in, const float* dout, int n, int k, int o, int silu_in, float* dW, float* db,
                               int accumulate, hipStream_t s);
// din[n][k] (+)= (sum_o dout[n][o] W[o][k]) * act'(in[n][k])
hipError_t launch_linear_bwd_in(const float* in, const float* dout, const float* W, int n, int k, int o, int silu_in,
                                float* din, int accumulate, hipStream_t s);

// ---------------------------------------------------------------- attention backward (attention_bwd.hip)
// Operand copies for the backward kernels, per (item, head):
//   "T" layout  [item][H][64][Tp]: head dim major, positions contiguous in the forward kernel's PV key order
//               (bits 2 <-> 3 of the position swapped inside every 16), zero tail for positions >= T
//   natural     [item][H][T][64]
// (mean != nullptr: mean[item*H + h][64] is subtracted from the valid positions -- centred copy)
hipError_t launch_attn_to_T(int dtype, const void* nat, int64_t item_stride, int64_t head_stride, int row_stride,
                            int n_items, int H, int T, int Tp, const float* mean, void* outT, hipStream_t s);
// mean[nh][64] over the T positions of a natural-layout tensor
hipError_t launch_attn_mean_nat(int dtype, const void* nat, int n_heads_total, int T, float* mean, hipStream_t s);
// natural copy of a T-layout tensor, CENTRED when vmean != nullptr: vmean[item][H][64] = mean over the positions is
// computed first and subtracted (attention_bwd.hip, "Precision")
// nat_lo (optional): 16-bit rounding residual of the centred values (hi + lo operand pair)
// the three means + the three centred copies of one attention backward in two launches (same bodies as the single launches below)
// vT_lo (optional): the forward's rounding residuals of v (hi + lo v operands): the mean and the centred pair are then formed from vT + vT_lo
hipError_t launch_attn_prep(int dtype, const void* q, const void* k, const void* vT, const void* vT_lo, int n_items, int H, int T, int Tp, float* qmean,
                            float* kmean, float* vmean, void* qT, void* kT, void* vnat, void* vnat_lo, hipStream_t s);
hipError_t launch_attn_from_T(int dtype, const void* inT, int n_items, int H, int T, int Tp, float* vmean, void* nat, void* nat_lo,
                              hipStream_t s);
struct AttnBwdArgs {
    const void* q; const void* k; const void* v;       // natural [item][H][T][64] (q pre-scaled by log2e/8, RoPE applied; v CENTRED)
    const void* vlo;                                   // rounding residual of the centred v (second operand of dP = dO v^T)
    const float* vmean;                                // [item][H][64]: what was subtracted from v
    unsigned* dsmax;                                   // [item][H]: bits of max |dS| bound (written by the dQ kernel, read by dK/dV)
    const void* qT; const void* kT; const void* dOT;   // "T" layout; qT and kT CENTRED
    const float* qmean; const float* kmean;            // [item][H][64]: what was subtracted from qT / kT
    const void* dO; int dO_row_stride;                 // time-major [item][T][H*64] (row stride H*64)
    const float* lse;                                  // [item][H][T]: log2-sum-exp of the forward
    unsigned* gmax;                                    // [3] bit patterns of max |dq|, max |dk|, max |dv| (atomicMax by the kernels; zeroed by the caller; may be null)
    float* alphaq;                                     // [item][H][T]: power-of-two operand scale of each query's dS column (first pass -> second pass)
    float* Dq; float* Fq; float* aq;                   // [item][H][T]: sum_k P f dP', sum_k P f, dO . vmean -- written by the dQ
                                                       // kernel (first pass), read by the dK/dV kernel
    const float* kbias; int mask_mod;                  // [mask_mod][Tp]
    const int* kv_end;
    float* dq; float* dk; float* dv;                   // natural [item][H][T][64] fp32: d(q_rope)*8, d(k_rope)/ln2, dv
    int T, Tp, H, n_items;
    DropCfg drop;
    const void* zeros;
};
hipError_t launch_attn_bwd_dq(int dtype, const AttnBwdArgs& a, hipStream_t s);
hipError_t launch_attn_bwd_dkv(int dtype, const AttnBwdArgs& a, hipStream_t s);
// Local power-of-two scales of the attention-input gradients.  d q and d k are ~1/T of d v in magnitude (P ~ 1/T), far
// below f16's normal range at the pass-wide gradient scale, so each of the three gets its own factor before it is
// rounded to 16 bits.  From the device-side maxima of |dq|, |dk|, |dv| (n values each) and the pass-wide pair gsc:
//   qs[0..1] = {f_c, 1 / f_c}        one common factor (max of the three -> ~2^8): operand of the fused q/k/v DGRAD GEMM,
//                                     whose fp32 result the consumer multiplies by qs[1]
//   qs[2..3], [4..5], [6..7]          {f_x gsc[0], 1 / (f_x gsc[0])} for x = q, k, v: unscale pairs of the three WGRAD outputs
//   qs[8..10]                         f_q, f_k, f_v (each tensor's own maximum -> ~2^8): operand of the WGRAD GEMM
hipError_t launch_qkv_grad_scales(const float* dq, const float* dk, const float* dv, int64_t n, const float* gsc,
                                  unsigned* bits3, float* qs, hipStream_t s, bool have_max = false);   // have_max: bits3 already holds the three maxima (attention backward kernels)
// RoPE^T on dq, dk, the 1/8 and ln2 factors, and packing into the time-major 16-bit operands [item][T][3*H*64]:
// d16 (common factor qs[0], dgrad) and w16 (per-tensor factors qs[8..10], wgrad)
hipError_t launch_qkv_grad_pack(int dtype, const float* dq, const float* dk, const float* dv, const float* rope_cos,
                                const float* rope_sin, int n_items, int H, int T, const float* qs, void* d16, void* w16,
                                hipStream_t s);

}  // namespace st
