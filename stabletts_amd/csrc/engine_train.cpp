// Training side of libstabletts_hip.so: a forward evaluation of the estimator that keeps the activations and the
// matching backward pass, both as launch sequences of hand-written gfx950 kernels.  Autograd counterpart of
// models/estimator.py:103-138 (Decoder.forward) + models/diffusion_transformer.py:98-121 as exercised by
// CFMDecoder.compute_loss (models/flow_matching.py:69-100) under DDP in train.py:78-81.
//
// Structure of the backward pass:
//   * data gradients of every convolution run through the FORWARD implicit-GEMM kernel with transposed, tap-flipped
//     weights (dX[t][ci] = sum_j sum_co W[co][ci][2-j] dY[t+j-1][co] is itself a k=3 convolution);
//   * weight gradients run through the same kernel too: with K-contiguous transposed copies of the activations and
//     of dY (train_kernels.hip) dW = dY^T X is the kernel's contraction with the frame axis as K, split over row
//     chunks and reduced deterministically;
//   * attention: flash-style recomputation (attention_bwd.hip);
//   * FiLM / LayerNorm+adaLN / gated residuals / SiLU+dropout: one frame per wave, per-(item, channel) sums kept in
//     registers and reduced without atomics;
//   * the per-item vectors (time MLP, FiLM and adaLN linears) are tiny fp32 kernels.
// The training forward uses plain GEMM epilogues + element-wise kernels (nothing fused across ops) because every
// intermediate is needed again; the fused inference path of engine.cpp is untouched.
#include "engine_internal.h"
#include "train_launch.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace st;

namespace sthost {

struct LayerAct {
    float *lscout, *x1, *o32, *x2, *f32b, *x3;
    void *h1, *h1lo, *q, *k, *vt, *vt_lo, *attn16, *h2, *a16, *u16, *x3_16;      // vt_lo: rounding residuals of v (ts->v_lo), else nullptr
    float* lse;
};

struct TrainState {
    // transposed (dgrad) weights
    Conv finT, inxT, incT, preT[3];
    std::vector<Conv> ffn1T, ffn2T, oprojT, qkvT, lscTa, lscTb;
    std::vector<Conv> qkv2;                     // ST_TRAIN_VLO=2: q/k/v weights over K = [h_hi | h_lo]: [W_q 0; W_k 0; W_v W_v]
    std::vector<void*> owned;
    // fp32 parameter gradients, reference shapes, in ONE flat buffer: parameter-name order (st_param_info's), every slice starting
    // on a 64-byte boundary (goff; st_train_grad_offset).  gbase = where the running backward writes them: the engine's own
    // grad_flat (st_train_backward + st_param_grad) or a buffer of the caller (st_train_backward_part: no staging copy).
    std::map<std::string, float*> grads;       // slices of grad_flat
    std::map<std::string, int64_t> goff;
    float* grad_flat = nullptr; int64_t grad_numel = 0;
    float* gbase = nullptr;
    int next_part = 0; int64_t part_serial = 0;   // st_train_backward_part: the part expected next of the backward of forward #part_serial
    bool own_grads_valid = false;                 // grad_flat holds the gradients of the LAST backward (false while / after a backward that wrote into a caller's buffer)
    // activation + scratch arena of the last train_forward
    char* ws = nullptr; size_t ws_cap = 0;
    int B = 0, T = 0, Tp = 0;
    float p_drop = 0.f; unsigned long long seed = 0;
    bool have_fwd = false;
    int64_t serial = 0;                        // of the forward whose activations are held (have_fwd); counts st_train_forward calls
    bool packed = false;                       // dgrad weights match the current parameters
    // saved tensors
    void *mu16, *x16, *x16lo, *a1, *p1, *a2, *p2, *cond16, *cond16lo, *h0_16, *x3lo;
    float *ada_pre, *dada_pre;                  // gin != hidden: output of adaLN_modulation.0 per block [L][N][C], and its gradient [N][C]
    float *maskbuf, *kbias, *cvec, *tvals, *emb, *th_pre, *tau, *film, *ada, *cpart, *h0, *v32;
    int *n_full, *kv_end;
    std::vector<LayerAct> L;
    // backward scratch
    float *dX, *dskip[8], *tmpC, *tmpF, *Dbuf, *Fbuf, *abuf, *alphabuf, *vmean, *qmean, *kmean, *dq, *dk, *dv, *partial, *part_b, *red, *dada, *dfilm, *dtau, *dth, *demb, *dcvec,
          *gin, *gsc;
    unsigned *gbits, *dsmax, *qbits;
    // one zeroed region per backward (a single memset in bwd_head): the maximum cells of every re-centring point (a group of
    // kMaxCellWords each), max |dq|, |dk|, |dv| per block (qbits_all + 4 i) and the dS bounds per block (dsmax_all + i N H)
    unsigned *zero_region = nullptr, *cells_ring = nullptr, *qbits_all = nullptr, *dsmax_all = nullptr; size_t zero_bytes = 0;
    int cell_idx = 0;
    float* gsc_ring = nullptr; int gsc_slots = 0;     // the pass-wide scale pair lives in a ring of slots: a re-centring point writes the NEXT slot (ts->gsc moves on)
    const float* skip_gsc[8] = {};                    // the slot each long-skip gradient was written under
    float* red_site[5] = {};                          // per-(item, chunk) partial sums of a block's five row kernels (one reduce launch per block)
    unsigned *drop_rowh_all = nullptr, *drop_colh_all = nullptr; size_t drop_row_stride = 0, drop_col_stride = 0;   // dropout tables of every attention site
    bool fuse_ln = true;                        // ST_FUSE_TRAIN_LN=0: stand-alone residual / LayerNorm kernels after out-proj and FFN conv_2 (A/B)
    bool fuse_silu = true;                      // ST_FUSE_SILU=0: stand-alone silu_drop / silu_bwd kernels (A/B and the bit-identity test)
    unsigned *drop_rowh, *drop_colh;            // dropout hash tables of the attention site being processed (launch_drop_tables)
    float* skip_sc;                             // {scale, 1 / scale} each long-skip gradient was written at
    float* qs;                                  // local scales of the attention-input gradients (launch_qkv_grad_scales)
    // 16-bit gradient operands.  Every tensor a weight-gradient GEMM reads has its OWN buffer: the weight gradients run on a side
    // stream (below) and may still be reading when the main chain produces the next operand.
    enum { DY_FFN2, DY_FFN1, DY_OPROJ, DY_DATTN, DY_QKVD, DY_QKVW, DY_LSC, DY_HEAD, DY_T0, DY_T1, DY_T2, DY_T3, DY_COUNT };
    void* dy[DY_COUNT] = {};
    void *vnat, *vnat_lo, *qT, *kT, *dOT, *xt, *dyt;
    // Side streams of the backward (ST_TRAIN_SIDE=0: everything on the caller's stream; results are bitwise identical either way):
    //   side  -- every weight-gradient GEMM + its reduction.  Nothing on the main chain reads their outputs; they only have to be
    //            finished when a backward part returns.  The main chain (data gradients, row kernels, attention) no longer waits
    //            for them, and their small reduce launches run beside the main chain's big kernels.
    //   side2 -- the gradient-INDEPENDENT operand copies of a block's attention backward (centred q / k / v copies, T layouts),
    //            forked at the start of the block and joined in front of its attention kernels.
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t ev_fork[16] = {}, ev_site[DY_COUNT] = {}, ev_join = nullptr, ev_blk = nullptr, ev_prep = nullptr;
    bool site_pending[DY_COUNT] = {};
    int fork_idx = 0;
    // The conv_q / conv_k weight gradients are ill-conditioned in v at random init (near-uniform softmax: dS ~ dO.(v_j - o_i), the
    // key-independent bulk of v cancels in that difference, its 16-bit error does not): with v and its input h1 as single 16-bit operands they
    // are 25 % off the fp32 oracle end to end at B = 4 x T = 1000 (cosine 0.991; tools/train_qk_split_estimate.py: rounding v 18 %, rounding
    // h1 11 %, rounding q and k 0.1 %).  Default (ST_TRAIN_VLO=2): the LayerNorm kernel also stores h1's rounding residuals, the q/k/v
    // GEMM runs over K = [h_hi | h_lo] against [W_q 0; W_k 0; W_v W_v] (q, k bit-identical) and stores v's residual plane, the attention
    // forward adds P v_lo, the backward's centred v pair is formed from both: 2.9 % (cosine 0.99957), 3.8 % at B = 64.  Costs 0.3 ms per
    // step (generic q/k/v tile -- twice the K for its v blocks only, GF_K2_V_ONLY -- instead of the weight-stationary kernel, 2x the PV MFMAs).  1: v as a pair but from h_hi only
    // (17 %); 0: rounds 1-5.
    bool v_lo = true, h_lo = true;
    bool use_side = true, side_prio = false;     // ST_TRAIN_SIDE=2: side streams at the device's lowest stream priority (0: no side streams)
    size_t partial_cap = 0, xt_cap = 0, dyt_cap = 0;
};

namespace {

inline ConvGemmArgs cargs(const st_engine* e, const Conv& cv, int n_items, int T, int B) {
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.w = cv.w; a.bias = cv.bias; a.cout = cv.cout; a.T = T; a.n_items = n_items;
    a.a0_mod = n_items; a.a1_mod = n_items; a.mask_mod = B;
    a.zeros = e->zeros;
    return a;
}

// dgrad weights of one forward convolution: rows = its input channels [ci_off, ci_off + ci_cnt) padded to cin_p,
// K = taps x (its output channels, padded to ld)
int pack_T(st_engine* e, TrainState* ts, Conv& cv, const std::string& wname, int cout, int cin_total, int taps, int ci_off,
           int ci_cnt, int cin_p, int ld, hipStream_t s) {
    cv.cout = cin_p; cv.cin = ld; cv.taps = taps; cv.split = false;
    const size_t bytes = (size_t)cin_p * taps * ld * 2;
    if (!cv.w) {        // the padding is zeroed once; a re-pack rewrites exactly the payload elements
        HIPCHK(e, hipMalloc(&cv.w, bytes)); ts->owned.push_back(cv.w);
        HIPCHK(e, hipMemsetAsync(cv.w, 0, bytes, s));
    }
    int rc = pk_weight_t(e, e->pk_T, P(e, wname), cout, cin_total, taps, ci_off, ci_cnt, cv.w, cin_p, ld, 0, s);
    if (rc) return rc;
    cv.bias = nullptr;
    return ST_OK;
}

}  // namespace

// Flat layout of the parameter gradients: name order, each slice aligned to 16 floats (vectorised optimizers, 64-byte rows).
int64_t train_grad_layout(const st_engine* e, std::map<std::string, int64_t>* offs) {
    int64_t off = 0;
    for (auto& kv : e->params) {
        if (offs) (*offs)[kv.first] = off;
        off += (kv.second.numel() + 15) / 16 * 16;
    }
    return off;
}

void train_invalidate(st_engine* e) {
    if (e->train) { e->train->packed = false; e->train->have_fwd = false; }
}

// (re)packs the transposed weights after a parameter update; lazy: inference-only users never pay for it
int train_prepare(st_engine* e, hipStream_t s) {
    if (e->kind != 0) return ST_OK;
    if (!e->train) {
        e->train = new TrainState();
        if (const char* v = getenv("ST_FUSE_SILU")) e->train->fuse_silu = atoi(v) != 0;
        if (const char* v = getenv("ST_FUSE_TRAIN_LN")) e->train->fuse_ln = atoi(v) != 0;
        if (const char* v = getenv("ST_TRAIN_VLO")) { e->train->v_lo = atoi(v) != 0; e->train->h_lo = atoi(v) == 2; }
        if (const char* v = getenv("ST_TRAIN_SIDE")) { e->train->use_side = atoi(v) != 0; e->train->side_prio = atoi(v) == 2; }
        TrainState* t0 = e->train;
        if (t0->use_side) {
            // ST_TRAIN_SIDE=2 (opt-in): the side streams at the device's LOWEST priority.  Alone in a process that is 0.7 % faster (18.64 ->
            // 18.51 ms per step, the main chain being the critical path) -- but inside bench.py, next to the inference engines' part streams,
            // the same setting took the step from 18.6 to 24.9 ms (streams alias onto the process's few hardware queues): not the default.
            int least = 0, greatest = 0;
            if (t0->side_prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) {
                HIPCHK(e, hipStreamCreateWithPriority(&t0->side, hipStreamNonBlocking, least));
                HIPCHK(e, hipStreamCreateWithPriority(&t0->side2, hipStreamNonBlocking, least));
            } else {
                HIPCHK(e, hipStreamCreateWithFlags(&t0->side, hipStreamNonBlocking));
                HIPCHK(e, hipStreamCreateWithFlags(&t0->side2, hipStreamNonBlocking));
            }
            for (auto& ev : t0->ev_fork) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            for (auto& ev : t0->ev_site) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIPCHK(e, hipEventCreateWithFlags(&t0->ev_join, hipEventDisableTiming));
            HIPCHK(e, hipEventCreateWithFlags(&t0->ev_blk, hipEventDisableTiming));
            HIPCHK(e, hipEventCreateWithFlags(&t0->ev_prep, hipEventDisableTiming));
        }
    }
    TrainState* ts = e->train;
    if (ts->packed) return ST_OK;
    ts->have_fwd = false;
    const int C = e->C, F = e->F, M = e->M, Mp = e->Mp, K = e->K, L = e->L;
    int rc;
    if (ts->grad_flat && pk_replay(e, e->pk_T, s, &rc)) {      // the usual case after an optimizer step: one launch
        if (rc) return rc;
        ts->packed = true;
        return ST_OK;
    }
    pk_begin(e->pk_T);
    if ((rc = pack_T(e, ts, ts->finT, "final_proj.weight", M, C, 1, 0, C, C, Mp, s))) return rc;
    if ((rc = pack_T(e, ts, ts->inxT, "in_proj.weight", C, C + M, 1, 0, M, Mp, C, s))) return rc;
    if ((rc = pack_T(e, ts, ts->incT, "in_proj.weight", C, C + M, 1, M, C, C, C, s))) return rc;
    if ((rc = pack_T(e, ts, ts->preT[0], "cond_proj.0.weight", F, M, K, 0, M, Mp, F, s))) return rc;
    if ((rc = pack_T(e, ts, ts->preT[1], "cond_proj.2.weight", F, F, K, 0, F, F, F, s))) return rc;
    if ((rc = pack_T(e, ts, ts->preT[2], "cond_proj.4.weight", C, F, K, 0, F, F, C, s))) return rc;
    ts->ffn1T.resize(L); ts->ffn2T.resize(L); ts->oprojT.resize(L); ts->qkvT.resize(L);
    ts->lscTa.resize(L / 2); ts->lscTb.resize(L / 2);
    for (int i = 0; i < L; ++i) {
        const std::string b = e->blk(i);
        if ((rc = pack_T(e, ts, ts->ffn1T[i], b + "mlp.conv_1.weight", F, C, K, 0, C, C, F, s))) return rc;
        if ((rc = pack_T(e, ts, ts->ffn2T[i], b + "mlp.conv_2.weight", C, F, K, 0, F, F, C, s))) return rc;
        if ((rc = pack_T(e, ts, ts->oprojT[i], b + "attn.conv_o.weight", C, C, 1, 0, C, C, C, s))) return rc;
        if (ts->h_lo) {      // forward q/k/v weights over K = [h_hi | h_lo]
            if ((int)ts->qkv2.size() != L) ts->qkv2.assign(L, Conv());
            Conv& q2 = ts->qkv2[i];
            q2.cout = 3 * C; q2.cin = 2 * C; q2.taps = 1; q2.bias = e->qkv[i].bias;
            if (!q2.w) { HIPCHK(e, hipMalloc(&q2.w, (size_t)3 * C * 2 * C * 2)); ts->owned.push_back(q2.w); HIPCHK(e, hipMemsetAsync(q2.w, 0, (size_t)3 * C * 2 * C * 2, s)); }
            int r2 = 0;
            for (const char* nm : {"q", "k", "v"}) {
                const float* w = P(e, b + "attn.conv_" + nm + ".weight");
                if ((rc = pk_weight(e, e->pk_T, w, C, C, 1, 0, C, q2.w, r2 * C, 2 * C, 0, r2 == 2 ? C : 2 * C, 0, s))) return rc;
                if (r2 == 2 && (rc = pk_weight(e, e->pk_T, w, C, C, 1, 0, C, q2.w, r2 * C, 2 * C, C, C, 0, s))) return rc;
                ++r2;
            }
        }
        // fused q/k/v: K of the dgrad GEMM = [dq | dk | dv] (3C)
        Conv& q = ts->qkvT[i];
        q.cout = C; q.cin = 3 * C; q.taps = 1;
        if (!q.w) { HIPCHK(e, hipMalloc(&q.w, (size_t)C * 3 * C * 2)); ts->owned.push_back(q.w); }
        int r = 0;
        for (const char* nm : {"q", "k", "v"}) {
            if ((rc = pk_weight_t(e, e->pk_T, P(e, b + "attn.conv_" + nm + ".weight"), C, C, 1, 0, C, q.w, C, 3 * C, r * C, s))) return rc;
            ++r;
        }
    }
    for (int j = 0; j < L / 2; ++j) {
        const std::string n = "lsc_layers." + std::to_string(j) + ".weight";
        if ((rc = pack_T(e, ts, ts->lscTa[j], n, C, 2 * C, K, 0, C, C, C, s))) return rc;
        if ((rc = pack_T(e, ts, ts->lscTb[j], n, C, 2 * C, K, C, C, C, C, s))) return rc;
    }
    if (!ts->grad_flat) {      // one allocation, parameter-name order (st_param_info's order): st_param_grads_flat copies it in one piece
        const int64_t total = train_grad_layout(e, &ts->goff);
        HIPCHK(e, hipMalloc((void**)&ts->grad_flat, (size_t)total * 4)); ts->owned.push_back(ts->grad_flat);
        HIPCHK(e, hipMemsetAsync(ts->grad_flat, 0, (size_t)total * 4, s));      // (the alignment gaps are never written)
        ts->grad_numel = total;
        for (auto& kv : ts->goff) ts->grads[kv.first] = ts->grad_flat + kv.second;
        ts->gbase = ts->grad_flat;
    }
    if ((rc = pk_end(e, e->pk_T, s))) return rc;
    ts->packed = true;
    return ST_OK;
}

int64_t train_bytes(const st_engine* e) {      // transposed weights + gradient buffers + activation / scratch arena
    if (!e->train) return 0;
    int64_t n = (int64_t)e->train->ws_cap;
    n += e->train->grad_numel * 4;
    auto add = [&](const Conv& c) { if (c.w) n += (int64_t)c.cout * c.taps * c.cin * 2; };
    add(e->train->finT); add(e->train->inxT); add(e->train->incT);
    for (auto& c : e->train->preT) add(c);
    for (auto* v : {&e->train->ffn1T, &e->train->ffn2T, &e->train->oprojT, &e->train->qkvT, &e->train->lscTa, &e->train->lscTb})
        for (auto& c : *v) add(c);
    return n;
}

void train_destroy(st_engine* e) {
    if (!e->train) return;
    for (void* p : e->train->owned) hipFree(p);
    if (e->train->side) { hipStreamSynchronize(e->train->side); hipStreamDestroy(e->train->side); }
    if (e->train->side2) { hipStreamSynchronize(e->train->side2); hipStreamDestroy(e->train->side2); }
    for (auto ev : e->train->ev_fork) if (ev) hipEventDestroy(ev);
    for (auto ev : e->train->ev_site) if (ev) hipEventDestroy(ev);
    for (auto ev : {e->train->ev_join, e->train->ev_blk, e->train->ev_prep}) if (ev) hipEventDestroy(ev);
    if (e->train->ws) hipFree(e->train->ws);
    delete e->train;
    e->train = nullptr;
}

namespace {

int layout_train(st_engine* e, TrainState* ts, int B, int T) {
    const int C = e->C, F = e->F, Mp = e->Mp, L = e->L, H = e->H, G = e->G;
    const int Tp = (T + 63) / 64 * 64;
    const size_t N = B, TT = T, R = N * TT;
    size_t off = 0;
    struct Slot { void** dst; size_t off; };
    std::vector<Slot> slots;
    auto want = [&](void** dst, size_t bytes) { slots.push_back({dst, off}); off = align_up(off + bytes, 256); };
    ts->L.assign(L, LayerAct());
    want(&ts->mu16, R * Mp * 2); want(&ts->x16, R * Mp * 2); want(&ts->x16lo, R * Mp * 2);
    want(&ts->a1, R * F * 2); want(&ts->p1, R * F * 2); want(&ts->a2, R * F * 2); want(&ts->p2, R * F * 2);
    want(&ts->cond16, R * C * 2); want(&ts->cond16lo, R * C * 2); want(&ts->h0_16, R * C * 2); want(&ts->x3lo, R * C * 2);
    want((void**)&ts->maskbuf, R * 4); want((void**)&ts->kbias, N * Tp * 4);
    want((void**)&ts->n_full, N * 4); want((void**)&ts->kv_end, N * 4);
    want((void**)&ts->cvec, N * G * 4); want((void**)&ts->tvals, N * 4);
    want((void**)&ts->ada_pre, (size_t)L * N * C * 4); want((void**)&ts->dada_pre, N * C * 4);
    want((void**)&ts->emb, N * C * 4); want((void**)&ts->th_pre, N * F * 4); want((void**)&ts->tau, N * C * 4);
    want((void**)&ts->film, (size_t)L * N * 2 * C * 4); want((void**)&ts->ada, (size_t)L * N * 6 * C * 4);
    want((void**)&ts->cpart, R * C * 4); want((void**)&ts->h0, R * C * 4); want((void**)&ts->v32, R * Mp * 4);
    for (int i = 0; i < L; ++i) {
        LayerAct& a = ts->L[i];
        if (i >= L / 2) want((void**)&a.lscout, R * C * 4); else a.lscout = nullptr;
        want((void**)&a.x1, R * C * 4); want((void**)&a.o32, R * C * 4); want((void**)&a.x2, R * C * 4);
        want((void**)&a.f32b, R * C * 4); want((void**)&a.x3, R * C * 4);
        want(&a.h1, R * C * 2); want(&a.q, R * C * 2); want(&a.k, R * C * 2); want(&a.vt, N * C * Tp * 2);
        if (ts->v_lo) want(&a.vt_lo, N * C * Tp * 2); else a.vt_lo = nullptr;
        if (ts->h_lo) want(&a.h1lo, R * C * 2); else a.h1lo = nullptr;
        want(&a.attn16, R * C * 2); want(&a.h2, R * C * 2); want(&a.a16, R * F * 2); want(&a.u16, R * F * 2);
        want(&a.x3_16, R * C * 2);
        want((void**)&a.lse, N * H * TT * 4);
    }
    // backward scratch
    want((void**)&ts->dX, R * C * 4);
    for (int j = 0; j < L / 2; ++j) want((void**)&ts->dskip[j], R * C * 4);
    want((void**)&ts->tmpC, R * C * 4); want((void**)&ts->tmpF, R * F * 4); want((void**)&ts->gin, R * Mp * 4);
    {
        const size_t w[TrainState::DY_COUNT] = {(size_t)C, (size_t)F, (size_t)C, (size_t)C, (size_t)3 * C, (size_t)3 * C, (size_t)C, (size_t)std::max(C, Mp),
                                                (size_t)C, (size_t)C, (size_t)F, (size_t)F};
        for (int k = 0; k < TrainState::DY_COUNT; ++k) want(&ts->dy[k], R * w[k] * 2);
    }
    want((void**)&ts->qs, (size_t)L * 64); want((void**)&ts->qbits, 16);      // qs: one 16-float record per block (the side stream's reduce reads it after the main chain has moved on)
    want((void**)&ts->drop_rowh, (N * H * TT + 64) * 4); want((void**)&ts->drop_colh, (size_t)(Tp / 2 + 64) * 4);
    want(&ts->vnat, R * C * 2); want(&ts->vnat_lo, R * C * 2); want((void**)&ts->dsmax, N * H * 4); want(&ts->qT, N * C * Tp * 2); want(&ts->kT, N * C * Tp * 2); want(&ts->dOT, N * C * Tp * 2);
    want((void**)&ts->Dbuf, N * H * TT * 4); want((void**)&ts->Fbuf, N * H * TT * 4); want((void**)&ts->abuf, N * H * TT * 4);
    want((void**)&ts->alphabuf, N * H * TT * 4);
    want((void**)&ts->vmean, N * H * 64 * 4); want((void**)&ts->qmean, N * H * 64 * 4); want((void**)&ts->kmean, N * H * 64 * 4);
    want((void**)&ts->dq, R * C * 4); want((void**)&ts->dk, R * C * 4); want((void**)&ts->dv, R * C * 4);
    // weight-gradient operands: the largest are (taps*Cin, Cout) = (3F, F) [cond_proj.2]; rows padded per split
    const size_t Rpad = align_up(R, 64) + 64 * 64;           // every split rounds its rows up to a multiple of 64
    ts->xt_cap = (size_t)3 * F * Rpad * 2; ts->dyt_cap = (size_t)F * Rpad * 2;
    want(&ts->xt, ts->xt_cap); want(&ts->dyt, ts->dyt_cap);
    ts->partial_cap = (size_t)64 * 3 * F * F * 4 / 4;        // S * taps*Cin * Cout fp32 with S chosen to fit (see wgrad())
    want((void**)&ts->partial, ts->partial_cap);
    want((void**)&ts->part_b, (Rpad / 64) * (size_t)std::max(F, 3 * C) * 4);
    want((void**)&ts->red, N * (size_t)red_chunks(T) * 2 * 256 * 4);
    for (int k = 0; k < 5; ++k) want((void**)&ts->red_site[k], N * (size_t)red_chunks(T) * 2 * 256 * 4);
    ts->gsc_slots = 4 * L + 8;
    want((void**)&ts->gsc_ring, (size_t)ts->gsc_slots * 16);
    {
        const size_t cells = (size_t)(4 * L + 8) * kMaxCellWords, qb = (size_t)4 * L, dm = (size_t)L * N * H;
        ts->zero_bytes = (cells + qb + dm) * 4;
        want((void**)&ts->zero_region, ts->zero_bytes);
    }
    ts->drop_row_stride = (size_t)N * H * TT + 64; ts->drop_col_stride = (size_t)(Tp / 2 + 64);
    want((void**)&ts->drop_rowh_all, (size_t)L * ts->drop_row_stride * 4); want((void**)&ts->drop_colh_all, (size_t)L * ts->drop_col_stride * 4);
    want((void**)&ts->dada, (size_t)L * N * 6 * C * 4); want((void**)&ts->dfilm, (size_t)L * N * 2 * C * 4);
    want((void**)&ts->dtau, N * C * 4); want((void**)&ts->dth, N * F * 4); want((void**)&ts->demb, N * C * 4);
    want((void**)&ts->dcvec, N * G * 4);
    want((void**)&ts->gsc, 16); want((void**)&ts->gbits, 16); want((void**)&ts->skip_sc, 64);
    if (off > ts->ws_cap) {
        if (ts->ws) { HIPCHK(e, hipDeviceSynchronize()); HIPCHK(e, hipFree(ts->ws)); ts->ws = nullptr; ts->ws_cap = 0; }
        HIPCHK(e, hipMalloc((void**)&ts->ws, off));
        ts->ws_cap = off;
    }
    for (auto& sl : slots) *sl.dst = ts->ws + sl.off;
    ts->cells_ring = ts->zero_region;
    ts->qbits_all = ts->zero_region + (size_t)(4 * L + 8) * kMaxCellWords;
    ts->dsmax_all = ts->qbits_all + (size_t)4 * L;
    ts->B = B; ts->T = T; ts->Tp = Tp;
    return ST_OK;
}

inline const float* xpre_of(const TrainState* ts, int i, int L) {
    return i == 0 ? ts->h0 : (i >= L / 2 ? ts->L[i].lscout : ts->L[i - 1].x3);
}

}  // namespace

}  // namespace sthost

using namespace sthost;

extern "C" {

int st_train_forward(st_engine* e, const float* t, const float* x, const float* mu, const float* mask, const float* c,
                     float* out, int B, int T, float p_dropout, uint64_t seed, void* stream) {
    int rc = check_ready(e, B, T); if (rc) return rc;
    if (e->kind != 0) return e->fail(ST_ERR_STATE, "this handle is not a CFM decoder (st_create)");
    if (!t || !x || !mu || !mask || !c || !out) return e->fail(ST_ERR_INVALID, "null tensor pointer");
    if (!(p_dropout >= 0.f && p_dropout < 1.f)) return e->fail(ST_ERR_INVALID, "p_dropout must be in [0, 1)");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    if ((rc = train_prepare(e, s))) return rc;
    TrainState* ts = e->train;
    ts->have_fwd = false;
    if ((rc = layout_train(e, ts, B, T))) return rc;
    if ((rc = ensure_rope(e, T, s))) return rc;
    ProfScope prof(e, s, PC_TRAIN_FWD, 0);
    const int C = e->C, F = e->F, Mp = e->Mp, L = e->L, H = e->H, N = B, Tp = ts->Tp;
    const int64_t R = (int64_t)N * T;
    ts->p_drop = p_dropout; ts->seed = seed;
    HIPCHK(e, launch_mask_prep(mask, B, T, Tp, ts->n_full, ts->kv_end, ts->kbias, nullptr, s));
    HIPCHK(e, launch_cvec_prep(mask, nullptr, B, T, ts->maskbuf, s));
    const float* m = ts->maskbuf;
    HIPCHK(e, launch_to_time_major(e->dt, mu, B, e->M, T, Mp, nullptr, ts->mu16, nullptr, s));
    HIPCHK(e, launch_to_time_major(e->dt, x, B, e->M, T, Mp, nullptr, ts->x16, ts->x16lo, s));
    HIPCHK(e, launch_cvec_prep(c, nullptr, B, e->G, ts->cvec, s));
    HIPCHK(e, hipMemcpyAsync(ts->tvals, t, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    // per-item vectors: time embedding -> MLP (pre-activation kept) -> FiLM; adaLN
    HIPCHK(e, launch_time_embed(ts->tvals, B, C, ts->emb, s));
    HIPCHK(e, launch_linear(ts->emb, B, C, P(e, "time_mlp.layer.0.weight"), P(e, "time_mlp.layer.0.bias"), F, ts->th_pre, 0, 0, s));
    HIPCHK(e, launch_linear(ts->th_pre, B, F, P(e, "time_mlp.layer.2.weight"), P(e, "time_mlp.layer.2.bias"), C, ts->tau, 1, 0, s));
    // FiLM (gamma, beta) and adaLN modulation of every block: per-item vectors, one launch per family and 8 blocks (were 2 L launches)
    for (int i0 = 0; i0 < L; i0 += 8) {
        LinearJobs jf; memset(&jf, 0, sizeof(jf));
        LinearJobs ja; memset(&ja, 0, sizeof(ja));
        for (int i = i0; i < std::min(L, i0 + 8); ++i) {
            const std::string pf = "blocks." + std::to_string(i) + ".time_fusion.film.";
            jf.in[jf.n] = ts->tau; jf.W[jf.n] = P(e, pf + "weight"); jf.bias[jf.n] = P(e, pf + "bias"); jf.out[jf.n] = ts->film + (size_t)i * N * 2 * C; jf.n += 1;
            const std::string pa = e->blk(i) + "adaLN_modulation.2.";
            const float* ain = ts->cvec;       // adaLN_modulation = [Linear(gin, hidden) if gin != hidden else Identity, SiLU, Linear(hidden, 6 hidden)]
            if (e->G != C) {                   // (diffusion_transformer.py:92-96)
                const std::string p0 = e->blk(i) + "adaLN_modulation.0.";
                float* pre = ts->ada_pre + (size_t)i * N * C;
                HIPCHK(e, launch_linear(ts->cvec, N, e->G, P(e, p0 + "weight"), P(e, p0 + "bias"), C, pre, 0, 0, s));
                ain = pre;
            }
            ja.in[ja.n] = ain; ja.W[ja.n] = P(e, pa + "weight"); ja.bias[ja.n] = P(e, pa + "bias"); ja.out[ja.n] = ts->ada + (size_t)i * N * 6 * C; ja.n += 1;
        }
        HIPCHK(e, launch_linear_multi(jf, B, C, 2 * C, 0, 0, s));
        HIPCHK(e, launch_linear_multi(ja, N, C, 6 * C, 1, 0, s));
    }
    const DropCfg nodrop = make_drop(0.f, 0, 0);
    if (make_drop(p_dropout, seed, 1).thresh16) {      // the hash tables of all L attention sites, once: forward and backward read them
        DropSeeds sd; memset(&sd, 0, sizeof(sd));
        if (L > 16) return e->fail(ST_ERR_INVALID, "training supports up to 16 blocks");
        for (int i = 0; i < L; ++i) sd.seed[i] = make_drop(p_dropout, seed, 2 * i + 1).seed;
        HIPCHK(e, launch_drop_tables_multi(sd, L, N * H * T + 64, Tp / 2, ts->drop_rowh_all, ts->drop_colh_all, ts->drop_row_stride, ts->drop_col_stride, s));
    }
    // cond prenet (estimator.py:83-89,118): pre-activations kept for SiLU'
    {
        // conv -> SiLU twice: the activation in the GEMM's epilogue where the phased kernel runs (as the FFN's, bit-identical)
        auto conv_silu = [&](ConvGemmArgs a, void* pre16, void* act16) -> int {
            a.out16 = pre16;
            if (ts->fuse_silu && gemm_is_phased(e, 3, a)) {
                a.act16 = act16;
                HIPCHK(e, gemm(e, 3, EPI_SILU, a, s));
            } else {
                HIPCHK(e, gemm(e, 3, EPI_F32, a, s));
                HIPCHK(e, launch_silu_drop(e->dt, pre16, act16, nullptr, 1, T, F, R, nodrop, s));
            }
            return ST_OK;
        };
        ConvGemmArgs a = cargs(e, e->pre[0], N, T, B); a.a0 = ts->mu16; a.c0 = Mp;
        if ((rc = conv_silu(a, ts->a1, ts->p1))) return rc;
        a = cargs(e, e->pre[1], N, T, B); a.a0 = ts->p1; a.c0 = F;
        if ((rc = conv_silu(a, ts->a2, ts->p2))) return rc;
        a = cargs(e, e->pre[2], N, T, B); a.a0 = ts->p2; a.c0 = F; a.out16 = ts->cond16; a.out16_lo = ts->cond16lo;
        HIPCHK(e, gemm(e, 3, EPI_F32, a, s));
        a = cargs(e, e->inc, N, T, B); a.a0 = ts->cond16; a.c0 = C; a.a1 = ts->cond16lo; a.c1 = C; a.c2 = C; a.out32 = ts->cpart;
        HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
    }
    {   // in_proj (estimator.py:120-121): output kept in fp32 (FiLM input / long skip) and 16 bit (long-skip operand)
        ConvGemmArgs a = cargs(e, e->inx, N, T, B);
        a.a0 = ts->x16; a.c0 = Mp; a.a1 = ts->x16lo; a.c1 = Mp; a.c2 = Mp; a.bias = nullptr;
        a.add32 = ts->cpart; a.add_clamp = N; a.out32 = ts->h0; a.out16 = ts->h0_16;
        HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
    }
    for (int i = 0; i < L; ++i) {
        LayerAct& A = ts->L[i];
        const float* ada_i = ts->ada + (size_t)i * N * 6 * C;
        if (i >= L / 2) {   // long-skip merge (estimator.py:131-132)
            const int j = i - L / 2;
            const int src = L - 1 - i;      // 2, 1, 0 -> x3[1], x3[0], h0
            ConvGemmArgs a = cargs(e, e->lsc[j], N, T, B);
            a.a0 = ts->L[i - 1].x3_16; a.c0 = C; a.a1 = src == 0 ? ts->h0_16 : ts->L[src - 1].x3_16; a.c1 = C; a.out32 = A.lscout;
            HIPCHK(e, gemm(e, 3, EPI_F32, a, s));
        }
        {   // FiLM * mask -> x1 ; LN1 + modulate -> h1
            TrainLnArgs a; memset(&a, 0, sizeof(a));
            a.xin = xpre_of(ts, i, L); a.xout = A.x1; a.h16 = A.h1; a.h16lo = A.h1lo;
            a.film = ts->film + (size_t)i * N * 2 * C; a.film_stride = 2 * C; a.film_mod = N;
            a.ada = ada_i; a.ada_stride = 6 * C; a.shift_off = 0; a.scale_off = C;
            a.mask = m; a.mask_mod = B; a.mask_out = 0; a.T = T; a.rows = (int)R;
            HIPCHK(e, launch_train_ln(e->dt, a, s));
        }
        {
            ConvGemmArgs a = cargs(e, ts->h_lo ? ts->qkv2[i] : e->qkv[i], N, T, B);
            a.a0 = A.h1; a.c0 = C; a.q = A.q; a.k = A.k; a.vt = A.vt; a.vt_lo = A.vt_lo; a.rope_cos = e->rope_cos; a.rope_sin = e->rope_sin;
            if (ts->h_lo) { a.a1 = A.h1lo; a.c1 = C; a.flags |= GF_K2_V_ONLY; }      // K = [h_hi | h_lo] against [W 0] (q, k rows: bit-identical) / [W_v W_v] (v rows)
            a.Tp = Tp; a.n_heads = H; a.qscale = 1.4426950408889634f / sqrtf((float)(C / H));
            if ((int)e->qkv_frag.size() == e->L) a.w_frag = e->qkv_frag[i];      // weight-stationary kernel on big batches (bit-identical; re-packed with the other forward weights)
            HIPCHK(e, gemm(e, 1, EPI_QKV, a, s));
        }
        {
            AttnArgs a; memset(&a, 0, sizeof(a));
            a.q = A.q; a.k = A.k; a.vt = A.vt; a.vt_lo = A.vt_lo; a.out = A.attn16; a.kbias = ts->kbias; a.mask_mod = B; a.zeros = e->zeros;
            a.kv_end = ts->kv_end; a.n_full = ts->n_full; a.T = T; a.Tp = Tp; a.H = H; a.n_items = N;
            a.lse = A.lse; a.drop = make_drop(p_dropout, seed, 2 * i + 1);
            if (a.drop.thresh16) { a.drop.rowh = ts->drop_rowh_all + (size_t)i * ts->drop_row_stride; a.drop.colh = ts->drop_colh_all + (size_t)i * ts->drop_col_stride; }
            HIPCHK(e, launch_attention(e->dt, a, s));
        }
        if (ts->fuse_ln) {   // o = (Wo attn + b) * mask ; x2 = x1 + g_msa * o ; LN2 + modulate, masked -> h2: one GEMM with the
            // inference epilogue (EPI_RESGATE + fused LayerNorm) that also stores the branch output o the gate's gradient needs
            ConvGemmArgs a = cargs(e, e->oproj[i], N, T, B);
            a.a0 = A.attn16; a.c0 = C; a.mask = m; a.gate = ada_i + 2 * C; a.gate_stride = 6 * C;
            a.res32 = A.x1; a.out32 = A.x2; a.branch32 = A.o32;
            a.ln_h16 = A.h2; a.ln_film = nullptr; a.ln_film_mod = 1;
            a.ln_ada = ada_i; a.ln_ada_stride = 6 * C; a.ln_shift_off = 3 * C; a.ln_scale_off = 4 * C; a.ln_mask_out = 1;
            HIPCHK(e, gemm(e, 1, EPI_RESGATE, a, s));
        } else {
            {   // o = (Wo attn + b) * mask
                ConvGemmArgs a = cargs(e, e->oproj[i], N, T, B);
                a.a0 = A.attn16; a.c0 = C; a.mask = m; a.flags = GF_MASK; a.out32 = A.o32;
                HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
            }
            {   // x2 = x1 + g_msa * o ; LN2 + modulate, masked -> h2
                TrainLnArgs a; memset(&a, 0, sizeof(a));
                a.xin = A.x1; a.xout = A.x2; a.h16 = A.h2;
                a.gate = ada_i + 2 * C; a.gate_stride = 6 * C; a.branch = A.o32;
                a.ada = ada_i; a.ada_stride = 6 * C; a.shift_off = 3 * C; a.scale_off = 4 * C;
                a.mask = m; a.mask_mod = B; a.mask_out = 1; a.T = T; a.rows = (int)R;
                HIPCHK(e, launch_train_ln(e->dt, a, s));
            }
        }
        {   // FFN (diffusion_transformer.py:25-30)
            ConvGemmArgs a = cargs(e, e->ffn1[i], N, T, B); a.a0 = A.h2; a.c0 = C; a.out16 = A.a16;
            const DropCfg dc = make_drop(p_dropout, seed, 2 * i);
            if (ts->fuse_silu && gemm_is_phased(e, 3, a)) {      // SiLU + dropout + mask in the GEMM's epilogue (bit-identical)
                a.act16 = A.u16; a.mask = m; a.drop_seed = dc.seed; a.drop_thresh16 = dc.thresh16; a.drop_scale = dc.scale;
                HIPCHK(e, gemm(e, 3, EPI_SILU, a, s));
            } else {
                HIPCHK(e, gemm(e, 3, EPI_F32, a, s));
                HIPCHK(e, launch_silu_drop(e->dt, A.a16, A.u16, m, B, T, F, R, dc, s));
            }
            a = cargs(e, e->ffn2[i], N, T, B); a.a0 = A.u16; a.c0 = F; a.mask = m;
            if (ts->fuse_ln) {      // f = (W2 u + b) * mask ; x3 = x2 + g_mlp * f (+ its 16-bit copies) in the GEMM's epilogue
                a.gate = ada_i + 5 * C; a.gate_stride = 6 * C; a.res32 = A.x2; a.out32 = A.x3; a.branch32 = A.f32b;
                a.out16 = A.x3_16; a.out16_lo = i + 1 == L ? ts->x3lo : nullptr;
                HIPCHK(e, gemm(e, 3, EPI_RESGATE, a, s));
            } else {
                a.flags = GF_MASK; a.out32 = A.f32b;
                HIPCHK(e, gemm(e, 3, EPI_F32, a, s));
            }
        }
        if (!ts->fuse_ln) {   // x3 = x2 + g_mlp * f  (+ 16-bit copies: long-skip / final_proj operands)
            TrainLnArgs a; memset(&a, 0, sizeof(a));
            a.xin = A.x2; a.xout = A.x3; a.x16 = A.x3_16; a.x16lo = i + 1 == L ? ts->x3lo : nullptr;
            a.gate = ada_i + 5 * C; a.gate_stride = 6 * C; a.branch = A.f32b;
            a.T = T; a.rows = (int)R;
            HIPCHK(e, launch_train_ln(e->dt, a, s));
        }
    }
    {
        ConvGemmArgs a = cargs(e, e->fin, N, T, B);
        a.a0 = ts->L[L - 1].x3_16; a.c0 = C; a.a1 = ts->x3lo; a.c1 = C; a.c2 = C; a.mask = m; a.flags = GF_MASK; a.out32 = ts->v32;
        HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
    }
    HIPCHK(e, launch_from_time_major(ts->v32, B, e->M, T, Mp, out, s));
    if (e->capture) {
        for (int i = 0; i < L; ++i) {
            const std::string bn = "t" + std::to_string(i) + ".";
            capture(e, bn + "x1", ts->L[i].x1, R * C, false, s); capture(e, bn + "x2", ts->L[i].x2, R * C, false, s);
            capture(e, bn + "x3", ts->L[i].x3, R * C, false, s); capture(e, bn + "h1", ts->L[i].h1, R * C, true, s);
            capture(e, bn + "q", ts->L[i].q, R * C, true, s); capture(e, bn + "k", ts->L[i].k, R * C, true, s);
            capture(e, bn + "vt", ts->L[i].vt, (int64_t)N * C * Tp, true, s); if (ts->L[i].vt_lo) capture(e, bn + "vtlo", ts->L[i].vt_lo, (int64_t)N * C * Tp, true, s);
             capture(e, bn + "attn", ts->L[i].attn16, R * C, true, s);
            capture(e, bn + "lse", ts->L[i].lse, (int64_t)N * H * T, false, s);
            capture(e, bn + "u", ts->L[i].u16, R * F, true, s);
        }
    }
    ts->have_fwd = true;
    ts->serial += 1;
    ts->gbase = ts->grad_flat; ts->next_part = 0;      // a pruned / abandoned multi-part backward of the previous forward leaves no state behind
    return ST_OK;
}

int64_t st_train_serial(const st_engine* e) {
    if (!e || !e->train || !e->train->have_fwd) return 0;
    return e->train->serial;
}

}  // extern "C"

namespace sthost {
namespace {

// dW / db of one convolution through the forward kernel.  X: up to two 16-bit sources [R][c0 | c1]; dY: [R][cout16]
// (16 bit, cout16 = padded channel count of the tensor); gradients go to the reference-layout fp32 tensors.
struct WgradOut { float* dW; int cin_total; int ci_off; int ci_cnt; int co_start; int co_cnt; float* db;
                  const float* unscale = nullptr; };     // {scale, 1 / scale} pair of the dY operand (default: the pass-wide ts->gsc)
int wgrad(st_engine* e, TrainState* ts, const void* x0, int c0, const void* x1, int c1, const void* dy, int cout16, int taps,
          const WgradOut* outs, int n_outs, hipStream_t s) {
    const int N = ts->B, T = ts->T;
    const int64_t R = (int64_t)N * T;
    const int cin = c0 + c1;
    const int frames = taps * cin;
    const int target_blocks = 512;        // transposed-copy path (cout = 128: final_proj): two rounds of its smaller blocks
    const int target_tn = 256;            // TN path: one block per CU
    if (cout16 % 256 == 0 && !(c0 & 63) && !(c1 & 63) && (!c1 || c0 % 256 == 0)) {
        // no transposed copies: the TN GEMM reads dY and X as they are (wgrad_tn.hip).  K (= items x 32-frame chunks) is split
        // into S ranges such that tiles x S fills ONE round of blocks (the kernel holds 128 KB of LDS: one block per CU) --
        // every block then carries the same share of the contraction and the reduce kernel reads the fewest planes
        const int tiles_tn = taps * ((cin + 255) / 256) * (cout16 / 256);
        const int kchunks = N * ((T + 31) / 32);
        const int S_want = std::max(1, std::min(std::min(kchunks, 64), target_tn / tiles_tn));      // <= 64 planes: the reduce kernel walks them serially
        int cps = (kchunks + S_want - 1) / S_want;
        while ((size_t)((kchunks + cps - 1) / cps) * frames * cout16 * 4 > ts->partial_cap && cps < kchunks) ++cps;
        const int S_tn = (kchunks + cps - 1) / cps;
        if ((size_t)S_tn * frames * cout16 * 4 <= ts->partial_cap) {
            bool need_b = false;
            for (int k = 0; k < n_outs; ++k) need_b = need_b || outs[k].db;
            // bias-gradient partials [S_tn][cout16] come out of the GEMM itself (the first N tile's blocks sum the dY fragments they hold)
            HIPCHK(e, launch_wgrad_tn(e->dt, dy, cout16, x0, c0, x1, c1, taps, N, T, cps, e->zeros, ts->partial, need_b ? ts->part_b : nullptr, s));
            WgradRed red[3];
            for (int k = 0; k < n_outs; ++k) {
                const WgradOut& o = outs[k];
                red[k] = {o.dW, o.db, o.unscale ? o.unscale : ts->gsc, o.cin_total, o.ci_off, o.ci_cnt, o.co_start, o.co_cnt};
            }
            HIPCHK(e, launch_wgrad_reduce_multi(ts->partial, need_b ? ts->part_b : nullptr, S_tn, cin, cout16, taps, red, n_outs, s));
            return ST_OK;
        }
    }
    // splits: enough blocks for the chip, bounded by the scratch capacities
    const int tiles = ((frames + 255) / 256) * std::max(1, cout16 / 256);
    int S = std::max(1, std::min(64, (target_blocks + tiles - 1) / tiles));
    S = (int)std::min<int64_t>(S, (R + 63) / 64);
    while (S > 1 && (size_t)S * frames * cout16 * 4 > ts->partial_cap) --S;
    const int Rs = (int)(((R + S - 1) / S + 63) / 64 * 64);
    if ((size_t)S * frames * Rs * 2 > ts->xt_cap || (size_t)S * cout16 * Rs * 2 > ts->dyt_cap ||
        (size_t)S * frames * cout16 * 4 > ts->partial_cap)
        return e->fail(ST_ERR_INVALID, "weight-gradient scratch too small");
    HIPCHK(e, launch_wgrad_xt(e->dt, x0, c0, x1, c1, N, T, taps, S, Rs, ts->xt, s));
    HIPCHK(e, launch_wgrad_dyt(e->dt, dy, cout16, R, S, Rs, ts->dyt, ts->part_b, s));
    ConvGemmArgs a; memset(&a, 0, sizeof(a));
    a.a0 = ts->xt; a.c0 = Rs; a.a0_mod = S; a.a1_mod = S;
    a.w = ts->dyt; a.w_item_stride = (long long)cout16 * Rs * 2;
    a.cout = cout16; a.T = frames; a.n_items = S; a.mask_mod = 1; a.zeros = e->zeros;
    a.out32 = ts->partial;
    HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
    for (int k = 0; k < n_outs; ++k) {
        const WgradOut& o = outs[k];
        const float* us = o.unscale ? o.unscale : ts->gsc;
        if (o.dW) HIPCHK(e, launch_wgrad_reduce(ts->partial, S, cin, cout16, taps, o.dW, o.cin_total, o.ci_off, o.ci_cnt, o.co_start, o.co_cnt, us, s));
        if (o.db) HIPCHK(e, launch_bias_reduce(ts->part_b, (int)((int64_t)S * Rs / 64), cout16, o.db, o.co_start, o.co_cnt, us, s));
    }
    return ST_OK;
}

float* G(TrainState* ts, const std::string& name) { return ts->gbase + ts->goff.at(name); }

// The weight gradient of one site on the side stream: it starts when everything enqueued on `s` so far (the producer of its dY
// operand) has run, and records the site's event when it is done with that operand.
int wgrad_side(st_engine* e, TrainState* ts, int site, const void* x0, int c0, const void* x1, int c1, int cout16, int taps,
               const WgradOut* outs, int n_outs, hipStream_t s) {
    const void* dy = ts->dy[site];
    if (!ts->use_side) return wgrad(e, ts, x0, c0, x1, c1, dy, cout16, taps, outs, n_outs, s);
    hipEvent_t f = ts->ev_fork[ts->fork_idx]; ts->fork_idx = (ts->fork_idx + 1) % 16;
    HIPCHK(e, hipEventRecord(f, s));
    HIPCHK(e, hipStreamWaitEvent(ts->side, f, 0));
    int rc = wgrad(e, ts, x0, c0, x1, c1, dy, cout16, taps, outs, n_outs, ts->side);
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(ts->ev_site[site], ts->side));
    ts->site_pending[site] = true;
    return ST_OK;
}
// ... and in front of the kernel that OVERWRITES a site's dY operand: wait until the previous block's weight gradient has read it
int site_free(st_engine* e, TrainState* ts, int site, hipStream_t s) {
    if (ts->use_side && ts->site_pending[site]) { HIPCHK(e, hipStreamWaitEvent(s, ts->ev_site[site], 0)); ts->site_pending[site] = false; }
    return ST_OK;
}
// end of a backward part: the parameter gradients must be complete when the call returns its stream to the caller
int side_join(st_engine* e, TrainState* ts, hipStream_t s) {
    if (!ts->use_side) return ST_OK;
    HIPCHK(e, hipEventRecord(ts->ev_join, ts->side));
    HIPCHK(e, hipStreamWaitEvent(s, ts->ev_join, 0));
    for (auto& p : ts->site_pending) p = false;
    return ST_OK;
}

}  // namespace
}  // namespace sthost

// The backward runs in three PARTS so that a data-parallel wrapper can start reducing the gradients of the layers that are done
// while the rest is still computing (train.py:49-51,78-81 under DDP: buckets become ready part by part):
//   part 0: d out -> final_proj, blocks L-1 .. L/2 with their long-skip convs        (parameters: final_proj, lsc_layers, blocks >= L/2)
//   part 1: blocks L/2-1 .. 0                                                        (parameters: blocks < L/2)
//   part 2: in_proj, the cond prenet, the time MLP; d x, d mu, d c                   (parameters: in_proj, cond_proj, time_mlp)
// Each block finishes its own per-item linears (adaLN modulation, FiLM) as soon as its d ada / d film rows are complete.
namespace sthost {
namespace {

struct BwdDims { int C, F, M, Mp, L, H, K, N, B, T, Tp; int64_t R; int chunks; };
BwdDims bwd_dims(const st_engine* e, const TrainState* ts) {
    BwdDims d{e->C, e->F, e->M, e->Mp, e->L, e->H, e->K, ts->B, ts->B, ts->T, ts->Tp, (int64_t)ts->B * ts->T, red_chunks(ts->T)};
    return d;
}

// Re-centres the pass-wide power-of-two gradient scale on the running gradient dX (device side, no host sync): whenever its
// maximum has left [2^1, 2^9) it is brought to [2^5, 2^6), dX is multiplied by the same factor (the kernel returns at once
// when it is 1) and every later fp32 result is un-scaled by the new pair.  Called at every point where the NEXT kernels round
// gradients to 16 bits: block boundaries and, inside a block, after each LayerNorm backward -- with trained-like weights (adaLN
// gates of O(1), peaky attention) the gradient grows by more than two orders of magnitude INSIDE a block and overflowed f16
// (65504) when the scale was only re-centred per block (profiles/r04_nan_trace.txt).
int recentre(st_engine* e, TrainState* ts, const BwdDims& d, bool have_max, hipStream_t s) {
    // have_max: the kernel that wrote dX last published max |dX| into next_cells(ts) (ln_bwd / add_rescaled); else an absmax pass runs here
    unsigned* cells = ts->cells_ring + (size_t)ts->cell_idx * kMaxCellWords;
    float* next = ts->gsc + 4;
    if (ts->cell_idx + 1 >= 4 * d.L + 8 || next >= ts->gsc_ring + (size_t)ts->gsc_slots * 4) return e->fail(ST_ERR_STATE, "re-centring ring exhausted");
    HIPCHK(e, launch_recentre(ts->dX, d.R * d.C, cells, have_max, ts->gsc, next, s));
    ts->gsc = next; ts->cell_idx += 1;
    return ST_OK;
}
inline unsigned* next_cells(TrainState* ts) { return ts->cells_ring + (size_t)ts->cell_idx * kMaxCellWords; }

int bwd_head(st_engine* e, TrainState* ts, const float* grad_out, hipStream_t s) {
    const BwdDims d = bwd_dims(e, ts);
    const int C = d.C, M = d.M, Mp = d.Mp, L = d.L, N = d.N, B = d.B, T = d.T;
    const int64_t R = d.R;
    const float* m = ts->maskbuf;
    int rc;
    const bool cap = e->capture;
    // ONE memset per backward: the maximum cells of the re-centring points, the attention kernels' max / bound cells of every block.
    // (d ada / d film rows, the per-item linears' gradients and d c / d tau are WRITTEN by their first producer: no zero fills.)
    HIPCHK(e, hipMemsetAsync(ts->zero_region, 0, ts->zero_bytes, s));
    ts->cell_idx = 0; ts->gsc = ts->gsc_ring;
    // d v (time-major, masked: out = (W x + b) * mask), as the 16-bit operand of the first GEMMs
    HIPCHK(e, launch_grad_scale(grad_out, (int64_t)B * M * T, ts->gbits, ts->gsc, s));
    HIPCHK(e, launch_to_time_major(e->dt, grad_out, B, M, T, Mp, ts->gin, nullptr, nullptr, s));
    HIPCHK(e, launch_cast16(e->dt, ts->gin, m, B, T, Mp, R, ts->gsc, ts->dy[TrainState::DY_HEAD], s));
    {   // final_proj
        WgradOut o = {G(ts, "final_proj.weight"), C, 0, C, 0, M, G(ts, "final_proj.bias")};
        if ((rc = wgrad_side(e, ts, TrainState::DY_HEAD, ts->L[L - 1].x3_16, C, nullptr, 0, Mp, 1, &o, 1, s))) return rc;
        ConvGemmArgs a = cargs(e, ts->finT, N, T, B); a.a0 = ts->dy[TrainState::DY_HEAD]; a.c0 = Mp; a.mask = m; a.flags = GF_MASK; a.out32 = ts->dX;
        HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
    }
    if (cap) { capture(e, "g.scale", ts->gsc, 2, false, s); capture(e, "g.x3_" + std::to_string(L - 1), ts->dX, R * C, false, s); }
    return ST_OK;
}

// The gradient-independent operand copies of block i's attention backward: V back to natural rows, centred, as a hi + lo pair;
// the means of q and k; centred T-layout copies of q and k (attention_bwd.hip's header says why).
int attn_prep(st_engine* e, TrainState* ts, int i, hipStream_t s) {
    const int H = e->H, N = ts->B, T = ts->T, Tp = ts->Tp;
    LayerAct& A = ts->L[i];
    HIPCHK(e, launch_attn_prep(e->dt, A.q, A.k, A.vt, A.vt_lo, N, H, T, Tp, ts->qmean, ts->kmean, ts->vmean, ts->qT, ts->kT, ts->vnat, ts->vnat_lo, s));
    return ST_OK;
}

int bwd_block(st_engine* e, TrainState* ts, int i, hipStream_t s) {
    const BwdDims d = bwd_dims(e, ts);
    const int C = d.C, F = d.F, L = d.L, H = d.H, K = d.K, N = d.N, B = d.B, T = d.T, Tp = d.Tp, chunks = d.chunks;
    const int64_t R = d.R;
    const float* m = ts->maskbuf;
    int rc;
    const bool cap = e->capture;
    if (ts->use_side) {      // (everything enqueued on s so far includes the previous block's attention kernels, the last readers of the copies)
        HIPCHK(e, hipEventRecord(ts->ev_blk, s));
        HIPCHK(e, hipStreamWaitEvent(ts->side2, ts->ev_blk, 0));
        if ((rc = attn_prep(e, ts, i, ts->side2))) return rc;
        HIPCHK(e, hipEventRecord(ts->ev_prep, ts->side2));
    }
    {
        LayerAct& A = ts->L[i];
        const std::string b = e->blk(i);
        const float* ada_i = ts->ada + (size_t)i * N * 6 * C;
        float* dada_i = ts->dada + (size_t)i * N * 6 * C;
        // The gradient's magnitude changes by orders of magnitude from block to block (FiLM's gamma multiplies the whole residual
        // stream: ~0.05 at initialisation) and, with trained-like weights, inside a block: the pass-wide power-of-two scale is
        // re-centred on the running gradient here and after each LayerNorm backward (recentre), so that every 16-bit operand
        // derived from it sits in f16's range.  Every fp32 result is un-scaled by the pair current at the time it is written; a
        // long-skip gradient remembers the scale it was written at (skip_sc) and is converted when it is added.
        // (block i + 1 wrote dX last through add_rescaled -- which published the maximum -- when it lies in the first half, else through a GEMM)
        if (i < L - 1 && (rc = recentre(e, ts, d, i + 1 < L / 2, s))) return rc;
        if (cap) capture(e, "g.scale_" + std::to_string(i), ts->gsc, 2, false, s);      // the scale block i's captured tensors carry
        RedSites sites; memset(&sites, 0, sizeof(sites));      // the block's per-(item, channel) sums: ONE reduce launch at its end
        auto site = [&](int k, int K, float* out, int out_stride, int off0, int off1) {
            sites.s[sites.n] = RedSite{ts->red_site[k], K, out, out_stride, {off0, off1}, ts->gsc};      // the pair current NOW un-scales these sums
            sites.n += 1;
            return ts->red_site[k];
        };
        // ---- x3 = x2 + g_mlp * f
        if ((rc = site_free(e, ts, TrainState::DY_FFN2, s))) return rc;
        HIPCHK(e, launch_gate_bwd(e->dt, ts->dX, A.f32b, ada_i + 5 * C, 6 * C, m, B, T, N, ts->dy[TrainState::DY_FFN2], site(0, 1, dada_i, 6 * C, 5 * C, 0), s));
        {   // conv_2
            WgradOut o = {G(ts, b + "mlp.conv_2.weight"), F, 0, F, 0, C, G(ts, b + "mlp.conv_2.bias")};
            if ((rc = wgrad_side(e, ts, TrainState::DY_FFN2, A.u16, F, nullptr, 0, C, K, &o, 1, s))) return rc;
            if ((rc = site_free(e, ts, TrainState::DY_FFN1, s))) return rc;
            ConvGemmArgs a = cargs(e, ts->ffn2T[i], N, T, B); a.a0 = ts->dy[TrainState::DY_FFN2]; a.c0 = C;
            const DropCfg dc = make_drop(ts->p_drop, ts->seed, 2 * i);
            if (ts->fuse_silu && K == 3 && gemm_is_phased(e, 3, a) && !a.bias) {      // d pre-activation straight from the dgrad's epilogue
                a.out16 = ts->dy[TrainState::DY_FFN1]; a.dact16 = A.a16; a.mask = m; a.drop_seed = dc.seed; a.drop_thresh16 = dc.thresh16; a.drop_scale = dc.scale;
                HIPCHK(e, gemm(e, K, EPI_SILU, a, s));
            } else {
                a.out32 = ts->tmpF;
                HIPCHK(e, gemm(e, K, EPI_F32, a, s));
                HIPCHK(e, launch_silu_bwd(e->dt, ts->tmpF, A.a16, m, B, T, F, R, dc, ts->dy[TrainState::DY_FFN1], s));
            }
        }
        {   // conv_1
            WgradOut o = {G(ts, b + "mlp.conv_1.weight"), C, 0, C, 0, F, G(ts, b + "mlp.conv_1.bias")};
            if ((rc = wgrad_side(e, ts, TrainState::DY_FFN1, A.h2, C, nullptr, 0, F, K, &o, 1, s))) return rc;
            ConvGemmArgs a = cargs(e, ts->ffn1T[i], N, T, B); a.a0 = ts->dy[TrainState::DY_FFN1]; a.c0 = F; a.out32 = ts->tmpC;
            HIPCHK(e, gemm(e, K, EPI_F32, a, s));
        }
        HIPCHK(e, launch_ln_bwd(A.x2, ts->tmpC, ada_i, 6 * C, 4 * C, m, B, 1, T, N, ts->dX, site(1, 2, dada_i, 6 * C, 4 * C, 3 * C), nullptr, next_cells(ts), s));
        if (cap) capture(e, "g.x2_" + std::to_string(i), ts->dX, R * C, false, s);
        if ((rc = recentre(e, ts, d, true, s))) return rc;
        if (cap) capture(e, "g.scale_a" + std::to_string(i), ts->gsc, 2, false, s);      // the scale of this block's attention-part tensors
        // ---- x2 = x1 + g_msa * o
        if ((rc = site_free(e, ts, TrainState::DY_OPROJ, s))) return rc;
        HIPCHK(e, launch_gate_bwd(e->dt, ts->dX, A.o32, ada_i + 2 * C, 6 * C, m, B, T, N, ts->dy[TrainState::DY_OPROJ], site(2, 1, dada_i, 6 * C, 2 * C, 0), s));
        {   // out projection
            WgradOut o = {G(ts, b + "attn.conv_o.weight"), C, 0, C, 0, C, G(ts, b + "attn.conv_o.bias")};
            if ((rc = wgrad_side(e, ts, TrainState::DY_OPROJ, A.attn16, C, nullptr, 0, C, 1, &o, 1, s))) return rc;
            ConvGemmArgs a = cargs(e, ts->oprojT[i], N, T, B); a.a0 = ts->dy[TrainState::DY_OPROJ]; a.c0 = C; a.out16 = ts->dy[TrainState::DY_DATTN];      // d attn (16 bit)
            HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
        }
        {   // attention
            if (ts->use_side) HIPCHK(e, hipStreamWaitEvent(s, ts->ev_prep, 0));      // the operand copies forked at the start of the block
            else if ((rc = attn_prep(e, ts, i, s))) return rc;
            HIPCHK(e, launch_attn_to_T(e->dt, ts->dy[TrainState::DY_DATTN], (int64_t)T * C, 64, C, N, H, T, Tp, nullptr, ts->dOT, s));
            AttnBwdArgs a; memset(&a, 0, sizeof(a));
            a.q = A.q; a.k = A.k; a.v = ts->vnat; a.vlo = ts->vnat_lo; a.dsmax = ts->dsmax; a.qT = ts->qT; a.kT = ts->kT; a.dOT = ts->dOT;
            a.dO = ts->dy[TrainState::DY_DATTN]; a.dO_row_stride = C; a.lse = A.lse; a.vmean = ts->vmean; a.qmean = ts->qmean; a.kmean = ts->kmean;
            a.Dq = ts->Dbuf; a.Fq = ts->Fbuf; a.aq = ts->abuf; a.alphaq = ts->alphabuf; a.kbias = ts->kbias; a.mask_mod = B;
            a.kv_end = ts->kv_end; a.dq = ts->dq; a.dk = ts->dk; a.dv = ts->dv; a.T = T; a.Tp = Tp; a.H = H; a.n_items = N;
            a.drop = make_drop(ts->p_drop, ts->seed, 2 * i + 1); a.zeros = e->zeros;
            if (a.drop.thresh16) { a.drop.rowh = ts->drop_rowh_all + (size_t)i * ts->drop_row_stride; a.drop.colh = ts->drop_colh_all + (size_t)i * ts->drop_col_stride; }
            unsigned* qbits = ts->qbits_all + 4 * i;      // (zeroed with the rest of zero_region in bwd_head)
            a.dsmax = ts->dsmax_all + (size_t)i * N * H;
            a.gmax = qbits;              // the kernels publish max |dq|, |dk|, |dv| themselves
            HIPCHK(e, launch_attn_bwd_dq(e->dt, a, s));
            HIPCHK(e, launch_attn_bwd_dkv(e->dt, a, s));
            // d q, d k are ~1/T of d v: each gets its own power-of-two factor before the rounding to 16 bits (f16's normal
            // range ends at 6e-5); the fused dgrad GEMM takes the copy with one common factor
            float* qs = ts->qs + 16 * i;
            HIPCHK(e, launch_qkv_grad_scales(ts->dq, ts->dk, ts->dv, R * C, ts->gsc, qbits, qs, s, true));
            if ((rc = site_free(e, ts, TrainState::DY_QKVW, s))) return rc;
            HIPCHK(e, launch_qkv_grad_pack(e->dt, ts->dq, ts->dk, ts->dv, e->rope_cos, e->rope_sin, N, H, T, qs, ts->dy[TrainState::DY_QKVD], ts->dy[TrainState::DY_QKVW], s));
        }
        if (cap) { capture(e, "g.dq_" + std::to_string(i), ts->dq, R * C, false, s); capture(e, "g.dk_" + std::to_string(i), ts->dk, R * C, false, s);
                   capture(e, "g.dv_" + std::to_string(i), ts->dv, R * C, false, s); capture(e, "g.dattn_" + std::to_string(i), ts->dy[TrainState::DY_DATTN], R * C, true, s); }
        {   // fused q/k/v projection
            WgradOut o[3];
            int r = 0;
            for (const char* nm : {"q", "k", "v"}) {
                o[r] = {G(ts, b + "attn.conv_" + nm + ".weight"), C, 0, C, r * C, C, G(ts, b + "attn.conv_" + nm + ".bias"), ts->qs + 16 * i + 2 + 2 * r};
                ++r;
            }
            if ((rc = wgrad_side(e, ts, TrainState::DY_QKVW, A.h1, C, nullptr, 0, 3 * C, 1, o, 3, s))) return rc;
            ConvGemmArgs a = cargs(e, ts->qkvT[i], N, T, B); a.a0 = ts->dy[TrainState::DY_QKVD]; a.c0 = 3 * C; a.out32 = ts->tmpC;
            HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
        }
        HIPCHK(e, launch_ln_bwd(A.x1, ts->tmpC, ada_i, 6 * C, C, m, B, 0, T, N, ts->dX, site(3, 2, dada_i, 6 * C, C, 0), ts->qs + 16 * i, next_cells(ts), s));
        if (cap) capture(e, "g.x1_" + std::to_string(i), ts->dX, R * C, false, s);
        if ((rc = recentre(e, ts, d, true, s))) return rc;
        if (cap) capture(e, "g.scale_b" + std::to_string(i), ts->gsc, 2, false, s);      // ... and of what follows (g.xin_i)
        // ---- x1 = (gamma * xpre + beta) * mask
        if (i >= L / 2 && (rc = site_free(e, ts, TrainState::DY_LSC, s))) return rc;
        HIPCHK(e, launch_film_bwd(e->dt, xpre_of(ts, i, L), ts->film + (size_t)i * N * 2 * C, 2 * C, N, m, B, T, N, ts->dX,
                                  i >= L / 2 ? ts->dy[TrainState::DY_LSC] : nullptr, site(4, 2, ts->dfilm + (size_t)i * N * 2 * C, 2 * C, 0, C), s));
        HIPCHK(e, launch_reduce_sites(sites, N, chunks, s));
        if (i >= L / 2) {   // long-skip conv: xpre_i = W [x3_{i-1} ; skip] + b
            const int j = i - L / 2, src = L - 1 - i;
            const std::string n = "lsc_layers." + std::to_string(j);
            const void* skip16 = src == 0 ? ts->h0_16 : ts->L[src - 1].x3_16;
            WgradOut o = {G(ts, n + ".weight"), 2 * C, 0, 2 * C, 0, C, G(ts, n + ".bias")};
            if ((rc = wgrad_side(e, ts, TrainState::DY_LSC, ts->L[i - 1].x3_16, C, skip16, C, C, K, &o, 1, s))) return rc;
            ConvGemmArgs a = cargs(e, ts->lscTb[j], N, T, B); a.a0 = ts->dy[TrainState::DY_LSC]; a.c0 = C; a.out32 = ts->dskip[src];
            ts->skip_gsc[src] = ts->gsc;      // the slot this long-skip gradient is written under (slots are never rewritten within a backward)
            HIPCHK(e, gemm(e, K, EPI_F32, a, s));
            a = cargs(e, ts->lscTa[j], N, T, B); a.a0 = ts->dy[TrainState::DY_LSC]; a.c0 = C; a.out32 = ts->dX;
            HIPCHK(e, gemm(e, K, EPI_F32, a, s));
        }
        // x3_{i-1} (or the in_proj output) is also a long-skip source of a later block: add that gradient
        if (i < L / 2) HIPCHK(e, launch_add_rescaled(ts->dX, ts->dskip[i], R * C, ts->gsc, ts->skip_gsc[i], i > 0 ? next_cells(ts) : nullptr, s));
        if (cap) capture(e, "g.xin_" + std::to_string(i), ts->dX, R * C, false, s);
    }
    {   // this block's per-item linears: adaLN modulation (-> d c), FiLM (-> d tau); its d ada / d film rows are complete now.
        // Every weight / bias gradient here has ONE producer (written, not accumulated: no zero fills); d c and d tau add up over
        // the blocks: the first block of a backward (L - 1) writes them, the others accumulate.
        if (e->G == C) return ST_OK;      // the usual case: the linears of all blocks of a backward part run batched at its end (bwd_linears)
        const int acc = i == L - 1 ? 0 : 1;
        const std::string pa = e->blk(i) + "adaLN_modulation.2.";
        float* gw = G(ts, pa + "weight"); float* gb = G(ts, pa + "bias");
        const float* dout = ts->dada + (size_t)i * N * 6 * C;
        if (e->G == C) {
            HIPCHK(e, launch_linear_bwd_w(ts->cvec, dout, N, C, 6 * C, 1, gw, gb, 0, s));
            HIPCHK(e, launch_linear_bwd_in(ts->cvec, dout, P(e, pa + "weight"), N, C, 6 * C, 1, ts->dcvec, acc, s));
        } else {        // through SiLU into adaLN_modulation.0, then into c
            const std::string p0 = e->blk(i) + "adaLN_modulation.0.";
            const float* pre = ts->ada_pre + (size_t)i * N * C;
            float* g0w = G(ts, p0 + "weight"); float* g0b = G(ts, p0 + "bias");
            HIPCHK(e, launch_linear_bwd_w(pre, dout, N, C, 6 * C, 1, gw, gb, 0, s));
            HIPCHK(e, launch_linear_bwd_in(pre, dout, P(e, pa + "weight"), N, C, 6 * C, 1, ts->dada_pre, 0, s));
            HIPCHK(e, launch_linear_bwd_w(ts->cvec, ts->dada_pre, N, e->G, C, 0, g0w, g0b, 0, s));
            HIPCHK(e, launch_linear_bwd_in(ts->cvec, ts->dada_pre, P(e, p0 + "weight"), N, e->G, C, 0, ts->dcvec, acc, s));
        }
        const std::string pf = "blocks." + std::to_string(i) + ".time_fusion.film.";
        gw = G(ts, pf + "weight"); gb = G(ts, pf + "bias");
        const float* dfo = ts->dfilm + (size_t)i * N * 2 * C;
        HIPCHK(e, launch_linear_bwd_w(ts->tau, dfo, N, C, 2 * C, 0, gw, gb, 0, s));
        HIPCHK(e, launch_linear_bwd_in(ts->tau, dfo, P(e, pf + "weight"), N, C, 2 * C, 0, ts->dtau, acc, s));
    }
    return ST_OK;
}

// The per-item linears (adaLN modulation -> d c, FiLM -> d tau) of blocks [lo, hi) in four launches: every block's d ada / d film rows
// are complete when its backward part ends.  `acc`: add to d c / d tau (the first part of a backward writes them).
int bwd_linears(st_engine* e, TrainState* ts, int lo, int hi, int acc, hipStream_t s) {
    const int C = e->C, N = ts->B;
    if (e->G != C) return ST_OK;          // (handled per block)
    for (int i0 = lo; i0 < hi; i0 += 8) {
        LinBwdJobs ja; memset(&ja, 0, sizeof(ja));
        LinBwdJobs jf; memset(&jf, 0, sizeof(jf));
        for (int i = i0; i < std::min(hi, i0 + 8); ++i) {
            const std::string pa = e->blk(i) + "adaLN_modulation.2.";
            ja.in[ja.n] = ts->cvec; ja.dout[ja.n] = ts->dada + (size_t)i * N * 6 * C; ja.W[ja.n] = P(e, pa + "weight");
            ja.dW[ja.n] = G(ts, pa + "weight"); ja.db[ja.n] = G(ts, pa + "bias"); ja.n += 1;
            const std::string pf = "blocks." + std::to_string(i) + ".time_fusion.film.";
            jf.in[jf.n] = ts->tau; jf.dout[jf.n] = ts->dfilm + (size_t)i * N * 2 * C; jf.W[jf.n] = P(e, pf + "weight");
            jf.dW[jf.n] = G(ts, pf + "weight"); jf.db[jf.n] = G(ts, pf + "bias"); jf.n += 1;
        }
        const int a = (acc || i0 > lo) ? 1 : 0;
        HIPCHK(e, launch_linear_bwd_w_multi(ja, N, C, 6 * C, 1, s));
        HIPCHK(e, launch_linear_bwd_in_multi(ja, ts->cvec, N, C, 6 * C, 1, ts->dcvec, a, s));
        HIPCHK(e, launch_linear_bwd_w_multi(jf, N, C, 2 * C, 0, s));
        HIPCHK(e, launch_linear_bwd_in_multi(jf, ts->tau, N, C, 2 * C, 0, ts->dtau, a, s));
    }
    return ST_OK;
}

int bwd_tail(st_engine* e, TrainState* ts, float* grad_x, float* grad_mu, float* grad_c, hipStream_t s) {
    const BwdDims d = bwd_dims(e, ts);
    const int C = d.C, F = d.F, M = d.M, Mp = d.Mp, K = d.K, N = d.N, B = d.B, T = d.T;
    const int64_t R = d.R;
    int rc;
    // ---- the time MLP's gradients and d c first: d tau and d c are complete since part 1, nothing below feeds them, and at the END of this
    // part these small launches queued behind the prenet's weight-gradient GEMMs on the side stream (120-210 us each in the kernel trace)
    HIPCHK(e, launch_linear_bwd_w(ts->th_pre, ts->dtau, N, F, C, 1, G(ts, "time_mlp.layer.2.weight"), G(ts, "time_mlp.layer.2.bias"), 0, s));
    HIPCHK(e, launch_linear_bwd_in(ts->th_pre, ts->dtau, P(e, "time_mlp.layer.2.weight"), N, F, C, 1, ts->dth, 0, s));
    HIPCHK(e, launch_linear_bwd_w(ts->emb, ts->dth, N, C, F, 0, G(ts, "time_mlp.layer.0.weight"), G(ts, "time_mlp.layer.0.bias"), 0, s));
    if (grad_c) HIPCHK(e, hipMemcpyAsync(grad_c, ts->dcvec, (size_t)N * e->G * 4, hipMemcpyDeviceToDevice, s));
    // ---- in_proj: h0 = Wx x + Wc cond + b
    void* const t0 = ts->dy[TrainState::DY_T0]; void* const t1 = ts->dy[TrainState::DY_T1]; void* const t2 = ts->dy[TrainState::DY_T2]; void* const t3 = ts->dy[TrainState::DY_T3];
    HIPCHK(e, launch_cast16(e->dt, ts->dX, nullptr, 1, T, C, R, nullptr, t0, s));
    {
        WgradOut ox = {G(ts, "in_proj.weight"), C + M, 0, M, 0, C, G(ts, "in_proj.bias")};
        if ((rc = wgrad_side(e, ts, TrainState::DY_T0, ts->x16, Mp, nullptr, 0, C, 1, &ox, 1, s))) return rc;
        WgradOut oc = {G(ts, "in_proj.weight"), C + M, M, C, 0, C, nullptr};
        if ((rc = wgrad_side(e, ts, TrainState::DY_T0, ts->cond16, C, nullptr, 0, C, 1, &oc, 1, s))) return rc;
        if (grad_x) {
            ConvGemmArgs a = cargs(e, ts->inxT, N, T, B); a.a0 = t0; a.c0 = C; a.out32 = ts->gin;
            HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
            HIPCHK(e, launch_unscale_inplace(ts->gin, R * Mp, ts->gsc, s));
            HIPCHK(e, launch_from_time_major(ts->gin, B, M, T, Mp, grad_x, s));
        }
        ConvGemmArgs a = cargs(e, ts->incT, N, T, B); a.a0 = t0; a.c0 = C; a.out16 = t1;     // d cond (16 bit)
        HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
    }
    {   // cond prenet
        const DropCfg nodrop = make_drop(0.f, 0, 0);
        WgradOut o2 = {G(ts, "cond_proj.4.weight"), F, 0, F, 0, C, G(ts, "cond_proj.4.bias")};
        if ((rc = wgrad_side(e, ts, TrainState::DY_T1, ts->p2, F, nullptr, 0, C, K, &o2, 1, s))) return rc;
        // d pre-activation = dgrad x SiLU': in the dgrad's epilogue where the phased kernel runs (as the FFN's)
        auto dgrad_silu = [&](ConvGemmArgs a, const void* pre16, void* dpre16) -> int {
            if (ts->fuse_silu && K == 3 && gemm_is_phased(e, 3, a) && !a.bias) {
                a.out16 = dpre16; a.dact16 = pre16;
                HIPCHK(e, gemm(e, K, EPI_SILU, a, s));
            } else {
                a.out32 = ts->tmpF;
                HIPCHK(e, gemm(e, K, EPI_F32, a, s));
                HIPCHK(e, launch_silu_bwd(e->dt, ts->tmpF, pre16, nullptr, 1, T, F, R, nodrop, dpre16, s));
            }
            return ST_OK;
        };
        ConvGemmArgs a = cargs(e, ts->preT[2], N, T, B); a.a0 = t1; a.c0 = C;
        if ((rc = dgrad_silu(a, ts->a2, t2))) return rc;
        WgradOut o1 = {G(ts, "cond_proj.2.weight"), F, 0, F, 0, F, G(ts, "cond_proj.2.bias")};
        if ((rc = wgrad_side(e, ts, TrainState::DY_T2, ts->p1, F, nullptr, 0, F, K, &o1, 1, s))) return rc;
        a = cargs(e, ts->preT[1], N, T, B); a.a0 = t2; a.c0 = F;
        if ((rc = dgrad_silu(a, ts->a1, t3))) return rc;
        WgradOut o0 = {G(ts, "cond_proj.0.weight"), M, 0, M, 0, F, G(ts, "cond_proj.0.bias")};
        if ((rc = wgrad_side(e, ts, TrainState::DY_T3, ts->mu16, Mp, nullptr, 0, F, K, &o0, 1, s))) return rc;
        if (grad_mu) {
            a = cargs(e, ts->preT[0], N, T, B); a.a0 = t3; a.c0 = F; a.out32 = ts->gin;
            HIPCHK(e, gemm(e, K, EPI_F32, a, s));
            HIPCHK(e, launch_unscale_inplace(ts->gin, R * Mp, ts->gsc, s));
            HIPCHK(e, launch_from_time_major(ts->gin, B, M, T, Mp, grad_mu, s));
        }
    }
    return ST_OK;
}

int bwd_check(st_engine* e, int64_t serial, int B_, int T_, const char* who) {
    if (e->kind != 0 || !e->train || !e->train->have_fwd)
        return e->fail(ST_ERR_STATE, std::string(who) + " needs a preceding st_train_forward (none held: never run, or invalidated by a parameter update)");
    if (serial != e->train->serial || B_ != e->train->B || T_ != e->train->T)
        return e->fail(ST_ERR_STATE, std::string(who) + ": the engine holds the activations of forward #" + std::to_string(e->train->serial) +
                       " (B=" + std::to_string(e->train->B) + ", T=" + std::to_string(e->train->T) + "), not of #" + std::to_string(serial) +
                       " (B=" + std::to_string(B_) + ", T=" + std::to_string(T_) + "): one backward per forward, before the next grad-enabled forward");
    return ST_OK;
}

int bwd_part(st_engine* e, TrainState* ts, int part, const float* grad_out, float* grad_x, float* grad_mu, float* grad_c, hipStream_t s) {
    ProfScope prof(e, s, PC_TRAIN_BWD, 0);
    const int L = e->L;
    int rc;
    if (part == 0) {
        if ((rc = bwd_head(e, ts, grad_out, s))) return rc;
        for (int i = L - 1; i >= L / 2; --i) if ((rc = bwd_block(e, ts, i, s))) return rc;
        if ((rc = bwd_linears(e, ts, L / 2, L, 0, s))) return rc;
    } else if (part == 1) {
        for (int i = L / 2 - 1; i >= 0; --i) if ((rc = bwd_block(e, ts, i, s))) return rc;
        if ((rc = bwd_linears(e, ts, 0, L / 2, 1, s))) return rc;
    } else {
        if ((rc = bwd_tail(e, ts, grad_x, grad_mu, grad_c, s))) return rc;
    }
    return side_join(e, ts, s);
}

}  // namespace
}  // namespace sthost

extern "C" {

int st_train_backward(st_engine* e, int64_t serial, int B_, int T_, const float* grad_out, float* grad_x, float* grad_mu,
                      float* grad_c, void* stream) {
    if (!e) return ST_ERR_INVALID;
    int rc = bwd_check(e, serial, B_, T_, "st_train_backward"); if (rc) return rc;
    if (!grad_out) return e->fail(ST_ERR_INVALID, "null tensor pointer");
    HIPCHK(e, hipSetDevice(e->device));
    TrainState* ts = e->train;
    ts->gbase = ts->grad_flat; ts->next_part = 0; ts->own_grads_valid = false;
    for (int part = 0; part < 3; ++part)
        if ((rc = bwd_part(e, ts, part, grad_out, grad_x, grad_mu, grad_c, (hipStream_t)stream))) return rc;
    ts->own_grads_valid = true;
    return ST_OK;
}

int st_train_backward_part(st_engine* e, int64_t serial, int B_, int T_, int part, const float* grad_out, float* grad_flat,
                           int64_t grad_numel, float* grad_x, float* grad_mu, float* grad_c, void* stream) {
    if (!e) return ST_ERR_INVALID;
    int rc = bwd_check(e, serial, B_, T_, "st_train_backward_part"); if (rc) return rc;
    TrainState* ts = e->train;
    if (part < 0 || part > 2) return e->fail(ST_ERR_INVALID, "st_train_backward_part: part must be 0, 1 or 2");
    if (part == 0) {
        if (!grad_out) return e->fail(ST_ERR_INVALID, "null tensor pointer");
        if (grad_flat && grad_numel != ts->grad_numel)
            return e->fail(ST_ERR_INVALID, "st_train_backward_part: grad_numel must be st_train_grad_numel()");
        ts->gbase = grad_flat ? grad_flat : ts->grad_flat;
        ts->part_serial = serial; ts->own_grads_valid = false;
    } else if (ts->next_part != part || ts->part_serial != serial) {
        return e->fail(ST_ERR_STATE, "st_train_backward_part: parts run in order 0, 1, 2 of ONE backward (expected part " +
                       std::to_string(ts->next_part) + ")");
    }
    HIPCHK(e, hipSetDevice(e->device));
    if ((rc = bwd_part(e, ts, part, grad_out, grad_x, grad_mu, grad_c, (hipStream_t)stream))) { ts->next_part = 0; return rc; }
    ts->next_part = part == 2 ? 0 : part + 1;
    if (part == 2) { ts->own_grads_valid = ts->gbase == ts->grad_flat; ts->gbase = ts->grad_flat; }
    return ST_OK;
}

int st_train_param_part(const st_engine* e, const char* name) {
    if (!e || !name || e->kind != 0) return ST_ERR_INVALID;
    const std::string n(name);
    if (!e->params.count(n)) return ST_ERR_INVALID;
    if (n.rfind("blocks.", 0) == 0) return atoi(n.c_str() + 7) >= e->L / 2 ? 0 : 1;
    if (n.rfind("final_proj.", 0) == 0 || n.rfind("lsc_layers.", 0) == 0) return 0;
    return 2;
}

int64_t st_train_grad_offset(const st_engine* e, const char* name) {
    if (!e || !name) return ST_ERR_INVALID;
    std::map<std::string, int64_t> offs;
    train_grad_layout(e, &offs);
    auto it = offs.find(name);
    return it == offs.end() ? (int64_t)ST_ERR_INVALID : it->second;
}

int64_t st_train_grad_numel(const st_engine* e) { return e ? train_grad_layout(e, nullptr) : (int64_t)ST_ERR_INVALID; }

int st_param_grads_flat(st_engine* e, float* dst, int64_t numel, void* stream) {
    if (!e || !dst) return ST_ERR_INVALID;
    if (!e->train || !e->train->grad_flat) return e->fail(ST_ERR_STATE, "no training state");
    if (numel != e->train->grad_numel) return e->fail(ST_ERR_INVALID, "st_param_grads_flat: numel must be the sum of all parameter sizes");
    if (!e->train->own_grads_valid) return e->fail(ST_ERR_STATE, "st_param_grads_flat: the last backward did not complete into the engine's own gradient buffer (it wrote into the caller's, or was abandoned)");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpyAsync(dst, e->train->grad_flat, (size_t)numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ST_OK;
}

int st_param_grad(st_engine* e, const char* name, float* dst, int64_t numel, void* stream) {
    if (!e || !name || !dst) return ST_ERR_INVALID;
    if (!e->train) return e->fail(ST_ERR_STATE, "no training state");
    auto it = e->train->grads.find(name);
    auto pit = e->params.find(name);
    if (it == e->train->grads.end() || pit == e->params.end()) return e->fail(ST_ERR_INVALID, std::string("unknown parameter: ") + name);
    if (pit->second.numel() != numel) return e->fail(ST_ERR_INVALID, std::string("size mismatch for gradient of ") + name);
    if (!e->train->own_grads_valid) return e->fail(ST_ERR_STATE, "st_param_grad: the last backward did not complete into the engine's own gradient buffer (it wrote into the caller's, or was abandoned)");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpyAsync(dst, it->second, (size_t)numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ST_OK;
}

}  // extern "C"
