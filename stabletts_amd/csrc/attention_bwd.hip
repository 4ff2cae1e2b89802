// Flash-attention backward for gfx950 (training path): recomputes P from q, k and the forward's log2-sum-exp,
// never materialises a T x T tensor.  Autograd counterpart of attention.hip, i.e. of
// F.scaled_dot_product_attention(q, k, v, attn_mask, dropout_p) at models/diffusion_transformer.py:77.
//
// Math per (item, head), scores in natural units S = q_r k_r^T / 8 (q_r, k_r post-RoPE), P = softmax(S + mask):
//     dV = Pd^T dO          Pd = P * keep / (1 - p)                       (dropout on the probabilities)
//     dP = (dO V^T) * keep / (1 - p)
//     dS = P * (dP - D)     D[q] = sum_d dO[q][d] O[q][d]
//     d q_r = dS k_r / 8 ,  d k_r = dS^T q_r / 8
// The forward stores q pre-scaled: q_s = q_r * log2(e)/8, so P = exp2(q_s k_r^T + bias - lse2) with lse2 the
// forward's log2-sum-exp.  The kernels accumulate dq_acc = dS k_r and dk_acc = dS^T q_s; the pack kernel applies
// d q_r = dq_acc / 8 and d k_r = dk_acc * ln 2, the inverse rotation of RoPE, and the 16-bit time-major packing.
//
// Precision: dS = P (dP - D) subtracts two numbers that share a large common part whenever V has a component that is
// constant over the keys (bias, adaLN shift): 16-bit rounding of dO, V and of the stored output O would then be
// amplified ~100x in dq / dk.  Two measures keep the backward at operand precision:
//   * V is centred: V' = V - c, c = mean of V over the positions (fp32 mean, then one 16-bit rounding of the small
//     centred value).  Without dropout dS is exactly invariant to c (sum_k P = 1); with dropout the extra term
//     P a (f - F), a = dO . c, F = sum_k P f restores exactness (f = dropout factor of the (query, key) pair).
//   * q and k have such components too, and the two accumulations dq = sum_k dS k, dk = sum_q dS q multiply them
//     by sums of dS that (nearly) cancel.  The 16-bit A operands K^T, Q^T are therefore centred as well, and the
//     mean part is added back in fp32 from fp32 row / column sums of dS: dq += kmean * sum_k dS, dk += qmean * sum_q dS.
//   * the centred V enters dP = dO V'^T as a hi + lo pair of 16-bit operands (two MFMAs): what is left of dP after the
//     subtraction of D is a small fraction of dP at weakly correlated V / K, so V's rounding would otherwise dominate;
//   * dS is rescaled by a power of two before it is rounded to 16 bits: its magnitude falls by orders of magnitude
//     from the last block to the first (f16 would go subnormal).  The dQ kernel bounds |dS| per query during its
//     first pass (each lane scales its own column), and publishes the per-(item, head) maximum for the dK/dV kernel;
//   * D is not taken from the stored 16-bit O: the dQ kernel makes a first pass over the keys that accumulates
//     D' = sum_k P f dP' and F with the SAME P and dP' the second pass uses, and publishes them for the dK/dV kernel.
//
// As in the forward kernel every MFMA is arranged so that no operand ever crosses lanes:
//   dQ kernel  (block = 256 queries, lane = query):   S^T = K Q^T and dP^T = V dO^T accumulate as [key][query]; their
//              element-wise product dS^T is the B operand of dQ^T[d][query] += K^T[d][key] dS^T[key][query].
//   dKV kernel (block = 128 keys, lane = key):        S = Q K^T and dP = dO V^T accumulate as [query][key]; P and dS
//              are the B operands of dV^T[d][key] += dO^T[d][query] P[query][key] and dK^T += Q^T dS.
// An accumulator used as a B operand presents its rows in the k-slot order "bits 2 <-> 3 swapped inside every 16";
// the matching A operands (K^T, Q^T, dO^T) therefore come from "T-layout" copies [item][H][64][Tp] that store the
// positions in exactly that order -- the layout the forward already uses for V^T (launch_attn_to_T builds them).
#include "common.h"
#include "train_launch.h"

#if defined(ST_ABL_NO_VLO) && !defined(ST_DEVTOOLS)
#error "ST_ABL_NO_VLO (ablation build: V' enters dP as ONE 16-bit operand) needs -DST_DEVTOOLS"
#endif
#ifdef ST_ABL_NO_VLO
constexpr bool kNoVlo = true;
#else
constexpr bool kNoVlo = false;
#endif

namespace st {

namespace {

constexpr int kTile = 64 * 128;      // one 64 x 64 16-bit tile, dense 128-B rows, XOR-swizzled like attention.hip
constexpr int kDkvLds = 8 * kTile + 5 * 2 * 64 * 4;      // dK/dV kernel: 4 tiles x 2 buffers + lse / D' / F / a / dropout row-hash rows

// natural-layout tile: 8 rows x 128 B per 1-KiB piece; rows >= T come from the zero page
__device__ __forceinline__ void dma_rows(const unsigned char* base, size_t row_stride, int row0, int T, const unsigned char* zeros,
                                         unsigned char* lds_tile, int piece, int lane) {
    const int row = piece * 8 + (lane >> 3);
    const int seg = (lane & 7) ^ ((row >> 1) & 7);
    const int pos = row0 + row;
    const unsigned char* src = pos < T ? base + (size_t)pos * row_stride + seg * 16 : zeros;
    glds16b(src, lds_tile + piece * 1024);
}
// T-layout tile: row = head dim, 64 consecutive (permuted) positions starting at col0, always inside Tp
__device__ __forceinline__ void dma_cols(const unsigned char* base, int Tp, int col0, unsigned char* lds_tile, int piece, int lane) {
    const int row = piece * 8 + (lane >> 3);
    const int seg = (lane & 7) ^ ((row >> 1) & 7);
    glds16b(base + ((size_t)row * Tp + col0 + seg * 8) * 2, lds_tile + piece * 1024);
}

template <class P>
__device__ __forceinline__ typename P::vec8 frag(const unsigned char* tile, int row_off, int swz, int slot) {
    return as_vec8<P>(*(const uint4*)(tile + row_off + ((slot ^ swz) << 4)));
}

// max |value| of the block's output tile into one cell (non-negative floats order like their bit patterns; NaN / inf do not
// set a scale, as in absmax_kernel): one atomic per wave -- replaces three 65-MB reading passes over dq, dk, dv per layer
__device__ __forceinline__ void publish_absmax(const f32x16_t (&o)[2], bool valid, unsigned* cell) {
    float m = 0.f;
    if (valid) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = fabsf(o[d][r]); if (v == v && v < 3.0e38f) m = fmaxf(m, v); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(cell, __float_as_uint(m));
}

// Augmented k-step that carries a row constant and a column constant THROUGH the score MFMA (as attention.hip does for its softmax
// reference): the row side holds -c as hi + mid + lo 16-bit terms (24 bits of c with f16 and with bf16 operands) and a 1, the column side
// three 1s and its own constant: the accumulator comes out as s - c_row + c_col with exact products -- no per-element add / subtract.
template <class P>
__device__ __forceinline__ typename P::vec8 aug_neg3(float c, int hi) {
    typename P::vec8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = to16<P>(0.f);
    const typename P::elem h = to16<P>(-c);
    const float r1 = -c - (float)h;
    const typename P::elem m = to16<P>(r1);
    const typename P::elem l = to16<P>(r1 - (float)m);
    if (hi == 0) { v[0] = h; v[1] = m; v[2] = l; v[3] = to16<P>(1.0f); }
    return v;
}
template <class P>
__device__ __forceinline__ typename P::vec8 aug_ones_plus(float c, int hi) {
    typename P::vec8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = to16<P>(0.f);
    if (hi == 0) { v[0] = to16<P>(1.0f); v[1] = to16<P>(1.0f); v[2] = to16<P>(1.0f); v[3] = to16<P>(c); }
    return v;
}

__device__ __forceinline__ void store_acc_rows(float* dst_row, const f32x16_t (&o)[2], int hi) {
    // lane = one position; o[d2][r] = value for head dim d2*32 + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
    for (int d2 = 0; d2 < 2; ++d2)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *(float4*)(dst_row + d2 * 32 + 8 * q4 + 4 * hi) =
                make_float4(o[d2][4 * q4 + 0], o[d2][4 * q4 + 1], o[d2][4 * q4 + 2], o[d2][4 * q4 + 3]);
}

}  // namespace

// ------------------------------------------------------------------------------------------ dQ
// PASS 1: only the first pass over the keys (D', F, the dS bound -> Dq / Fq / aq / alphaq and the per-(item, head) maximum);
// no dQ accumulators, no K^T tiles: 128 VGPRs, two 8-wave blocks per CU.  PASS 2: only the second pass (dS, dQ), with D', F and
// the column scale read back.  (One kernel doing both passes ran the cheap first pass at the second pass's 2 waves per SIMD.)
template <class P, bool DROP, int PASS>
__global__ __launch_bounds__(512, PASS == 1 ? 4 : 2) void attn_bwd_dq_kernel(const AttnBwdArgs a) {
    constexpr int NW = 8, QB = 32 * NW;
    using vec8 = typename P::vec8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(PASS == 1 ? 6 : 8) * kTile];
    unsigned char* Ks = smem;                 // K natural, 2 buffers
    unsigned char* Vs = smem + 2 * kTile;     // V' natural (hi)
    unsigned char* VLs = smem + 4 * kTile;    // V' natural (lo)
    unsigned char* KTs = smem + 6 * kTile;    // K'^T (T layout)

    const int T = a.T, Tp = a.Tp, H = a.H;
    const int qtiles = (T + QB - 1) / QB;
    const int total = a.n_items * H * qtiles;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int qt = lin % qtiles;
    const int nh = lin / qtiles;
    const int n = nh / H, h = nh % H;
    const int mb = n % a.mask_mod;
    const int kvend = a.kv_end[mb];
    const float* kbias = a.kbias + (size_t)mb * Tp;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int query = qt * QB + wave * 32 + l31;
    const bool qok = query < T;
    if (qt * QB >= kvend) {
        // Ragged batch: every query of this tile lies past the item's last valid frame.  d attn is exactly 0 there (the
        // out-projection's output is multiplied by the mask, diffusion_transformer.py:111), so dq = 0, D' = 0: no key loop.
        if (qok) {
            if constexpr (PASS == 2) {
                float* dst = a.dq + ((size_t)nh * T + query) * 64 + hi * 32;
#pragma unroll
                for (int i = 0; i < 8; ++i) *(float4*)(dst + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (hi == 0) {
                a.Dq[(size_t)nh * T + query] = 0.f; a.Fq[(size_t)nh * T + query] = 1.f; a.aq[(size_t)nh * T + query] = 0.f;
                a.alphaq[(size_t)nh * T + query] = 1.f;
            }
        }
        return;
    }

    const unsigned char* qbase = (const unsigned char*)a.q + ((size_t)nh * T) * 128;
    const unsigned char* kbase = (const unsigned char*)a.k + ((size_t)nh * T) * 128;
    const unsigned char* vbase = (const unsigned char*)a.v + ((size_t)nh * T) * 128;
    const unsigned char* vlbase = (const unsigned char*)a.vlo + ((size_t)nh * T) * 128;
    const unsigned char* ktbase = (const unsigned char*)a.kT + ((size_t)nh * 64) * Tp * 2;
    const unsigned char* dobase = (const unsigned char*)a.dO + ((size_t)n * T) * a.dO_row_stride * 2 + (size_t)h * 128;
    const unsigned char* zeros = (const unsigned char*)a.zeros;

    vec8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
        if (qok) {
            v = *(const uint4*)(qbase + (size_t)query * 128 + ks * 32 + hi * 16);
            w = *(const uint4*)(dobase + (size_t)query * a.dO_row_stride * 2 + ks * 32 + hi * 16);
        }
        qf[ks] = as_vec8<P>(v); dof[ks] = as_vec8<P>(w);
    }
    const float lse_q = qok ? a.lse[(size_t)nh * T + query] : 0.f;      // (carrying bias - lse through an augmented k-step as the dK/dV kernel does was
                                                                        //  SLOWER here, 207 -> 220 us: the key-side operand has to be rebuilt per tile from a global load)
    // a = dO[q] . c (c = the mean that was subtracted from V): this lane holds 32 of the 64 head dims, lane^32 the others
    float a_q = 0.f;
    {
        const float* cm = a.vmean + (size_t)nh * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) a_q += (float)dof[ks][e2] * cm[ks * 16 + hi * 8 + e2];
        a_q = xor32_sum(a_q);
    }
    constexpr bool dropping = DROP;
    const unsigned drop_rh = DROP ? a.drop.rowh[(size_t)nh * T + (qok ? query : T - 1)] : 0u;

    const int ntiles = (kvend + 63) >> 6;
    auto sgpr_ptr = [](const unsigned char* p) {
        const uintptr_t v = (uintptr_t)p;
        return (const unsigned char*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
    };
    auto issue = [&](int kt, int buf) {     // 8 pieces per tile, one per wave
        if constexpr (PASS == 1) {
            // SGPR base + one 32-bit per-lane offset for the three tiles (128 VGPRs); rows >= T are clamped to row T-1, a
            // valid row whose probability the key bias sends to 0 (those tiles are never inside the bias-free leading run)
            const int row = wave * 8 + (lane >> 3);
            const int seg = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned voff = (unsigned)(min(kt * 64 + row, T - 1) * 128 + seg * 16);
            glds16s(sgpr_ptr(kbase), voff, Ks + buf * kTile + wave * 1024);
            glds16s(sgpr_ptr(vbase), voff, Vs + buf * kTile + wave * 1024);
            glds16s(sgpr_ptr(vlbase), voff, VLs + buf * kTile + wave * 1024);
            return;
        }
        dma_rows(kbase, 128, kt * 64, T, zeros, Ks + buf * kTile, wave, lane);
        dma_rows(vbase, 128, kt * 64, T, zeros, Vs + buf * kTile, wave, lane);
        dma_rows(vlbase, 128, kt * 64, T, zeros, VLs + buf * kTile, wave, lane);
        if constexpr (PASS == 2) dma_cols(ktbase, Tp, kt * 64, KTs + buf * kTile, wave, lane);
    };
    int row_off[2], swz[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) { const int row = b * 32 + l31; row_off[b] = row * 128; swz[b] = (row >> 1) & 7; }

    f32x16_t o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float Dacc = 0.f, Facc = 0.f, D_q = 0.f, F_q = 1.f, rs = 0.f;       // rs = sum_k dS[q][k] (fp32)
    float mxpd = 0.f, mxp = 0.f, alpha = 1.f;      // bounds for |dS| (first pass) -> power-of-two operand scale of this lane's column
    if constexpr (PASS == 2) {
        if (qok) { D_q = a.Dq[(size_t)nh * T + query]; F_q = a.Fq[(size_t)nh * T + query]; alpha = a.alphaq[(size_t)nh * T + query]; }
    }

    const float a_al = a_q * alpha;                                          // (second pass only: alpha, D', F are final there)
    const float D2_al = (dropping ? D_q + a_q * F_q : D_q) * alpha;
    if (ntiles > 0) issue(0, 0);
    ST_DMA_WAIT(0);
    __syncthreads();
    // pass 0 over the key tiles: D' and F;  pass 1: dS and dQ
    for (int it = 0; it < ntiles; ++it) {
        constexpr bool second = PASS == 2;
        const int kt = it;
        const int buf = it & 1;
        if (it + 1 < ntiles) issue(it + 1, buf ^ 1);
        f32x16_t s[2], dp[2];
        vec8 dsf[4];
        auto elems = [&](int kb) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bz = *(const float4*)(kbias + kt * 64 + kb * 32 + 8 * g4 + 4 * hi);
                const float bzv[4] = {bz.x, bz.y, bz.z, bz.w};
                float fv4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (DROP) {     // keys 8 g4 + 4 hi + {0, 1} and + {2, 3}: two pair hashes (DropCfg, launch.h)
                    const int key0 = kt * 64 + kb * 32 + 8 * g4 + 4 * hi;
                    const uint2 ch = *(const uint2*)(a.drop.colh + (key0 >> 1));
                    const float2 f01 = drop_factors2(a.drop, drop_pair(drop_rh, ch.x)), f23 = drop_factors2(a.drop, drop_pair(drop_rh, ch.y));
                    fv4[0] = f01.x; fv4[1] = f01.y; fv4[2] = f23.x; fv4[3] = f23.y;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const float p = __builtin_amdgcn_exp2f(s[kb][r] + bzv[e] - lse_q);
                    const float f = fv4[e];
                    if (!second) {
                        Dacc += p * f * dp[kb][r]; Facc += p * f;
                        mxpd = fmaxf(mxpd, fabsf(p * f * dp[kb][r])); mxp = fmaxf(mxp, p);
                    } else {
                        // alpha dS = p f (dP' + a) alpha - p (D' + a F) alpha  (= alpha [p (f dP' - D') + p a (f - F)], the per-query constants folded
                        // once per lane: 5 instead of 7 vector instructions per element; rs accumulates the SCALED values)
                        float ds;
                        if (dropping) ds = (p * f) * fmaf(dp[kb][r], alpha, a_al) - p * D2_al;
                        else ds = p * fmaf(dp[kb][r], alpha, -D2_al);
                        rs += ds;
                        dsf[kb * 2 + (r >> 3)][r & 7] = to16<P>(ds);
                    }
                }
            }
        };
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[kb][r] = 0.f; dp[kb][r] = 0.f; }
            if constexpr (PASS == 2) {
                // Second pass (229 VGPRs of 256): the fragment reads of two k-steps are issued as ONE batch in front of their six MFMAs --
                // left to itself hipcc emits read -> s_waitcnt lgkmcnt(0) -> MFMA, an exposed LDS round trip per MFMA with two waves per
                // SIMD to hide it.  Same MFMAs in the same order: bit-identical.  (The first pass has 128 registers and the dK/dV kernel
                // 250: the same batching spills there.)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    vec8 fk[2], fv[2], fl[2];
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int ks = kh * 2 + k2;
                        fk[k2] = frag<P>(Ks + buf * kTile, row_off[kb], swz[kb], ks * 2 + hi);
                        fv[k2] = frag<P>(Vs + buf * kTile, row_off[kb], swz[kb], ks * 2 + hi);
                        if constexpr (!kNoVlo) fl[k2] = frag<P>(VLs + buf * kTile, row_off[kb], swz[kb], ks * 2 + hi);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int ks = kh * 2 + k2;
                        s[kb] = P::mfma(fk[k2], qf[ks], s[kb]);
                        dp[kb] = P::mfma(fv[k2], dof[ks], dp[kb]);
                        if constexpr (!kNoVlo) dp[kb] = P::mfma(fl[k2], dof[ks], dp[kb]);
                    }
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s[kb] = P::mfma(frag<P>(Ks + buf * kTile, row_off[kb], swz[kb], ks * 2 + hi), qf[ks], s[kb]);
                dp[kb] = P::mfma(frag<P>(Vs + buf * kTile, row_off[kb], swz[kb], ks * 2 + hi), dof[ks], dp[kb]);
                if constexpr (!kNoVlo) dp[kb] = P::mfma(frag<P>(VLs + buf * kTile, row_off[kb], swz[kb], ks * 2 + hi), dof[ks], dp[kb]);
            }
            }
            if constexpr (PASS == 1) { __builtin_amdgcn_sched_barrier(0); elems(kb); __builtin_amdgcn_sched_barrier(0); }   // one key block live at a time: 128 VGPRs
        }
        if constexpr (PASS == 2) { elems(0); elems(1); }
        if constexpr (!second) {
            if (it == ntiles - 1) {
                D_q = xor32_sum(Dacc); F_q = xor32_sum(Facc);
                // |dS| <= max p|f dP| + max p (|D| + |a| (1/(1-p) + F)): scale this query's column to ~2^10
                float bound = xor32_max(mxpd) + xor32_max(mxp) * (fabsf(D_q) + (dropping ? fabsf(a_q) * (a.drop.scale + F_q) : 0.f));
                if (!(bound > 0.f) || !qok) bound = 0.f;
                if (bound > 0.f) {
                    int ex = 10 - (int)ceilf(log2f(bound));
                    ex = ex < -40 ? -40 : (ex > 80 ? 80 : ex);
                    alpha = exp2f((float)ex);
                }
                // per-(item, head) maximum for the dK/dV kernel (order-independent: max of non-negative floats via their bits)
                float wmax = bound;
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
                if (lane == 0) atomicMax(a.dsmax + nh, __float_as_uint(wmax));
            }
        } else {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                vec8 ft[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) ft[g] = frag<P>(KTs + buf * kTile, row_off[d], swz[d], g * 2 + hi);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) o[d] = P::mfma(ft[g], dsf[g], o[d]);
            }
        }
        ST_DMA_WAIT(0);
        __syncthreads();
    }
    if constexpr (PASS == 1) {
        if (qok && hi == 0) {
            a.Dq[(size_t)nh * T + query] = D_q; a.Fq[(size_t)nh * T + query] = F_q; a.aq[(size_t)nh * T + query] = a_q;
            a.alphaq[(size_t)nh * T + query] = alpha;
        }
        return;
    }
    {   // undo the operand scale; K^T was centred: add kmean[d] * sum_k dS back (fp32)
        const float inv_alpha = 1.0f / alpha;
        rs = xor32_sum(rs) * inv_alpha;        // (the row sum was accumulated in scaled units; alpha is a power of two)
        const float* km = a.kmean + (size_t)nh * 64;
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d2][r] = o[d2][r] * inv_alpha + km[d2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] * rs;
    }
    if (qok) store_acc_rows(a.dq + ((size_t)nh * T + query) * 64, o, hi);
    if (a.gmax) publish_absmax(o, qok, a.gmax);
}

// ------------------------------------------------------------------------------------------ dK, dV
template <class P, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnBwdArgs a) {
    constexpr int NW = 4, KB = 32 * NW;
    using vec8 = typename P::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // kDkvLds bytes
    unsigned char* Qs = smem;                  // Q natural, 2 buffers
    unsigned char* dOs = smem + 2 * kTile;     // dO natural (time-major rows)
    unsigned char* QTs = smem + 4 * kTile;     // Q^T  (T layout)
    unsigned char* dOTs = smem + 6 * kTile;    // dO^T (T layout)
    float* lse_t = (float*)(smem + 8 * kTile);         // [2][64] each
    float* D_t = lse_t + 2 * 64;
    float* F_t = D_t + 2 * 64;
    float* a_t = F_t + 2 * 64;
    unsigned* rh_t = (unsigned*)(a_t + 2 * 64);     // dropout: rowh of the tile's queries (the same 64 values for every lane)
    constexpr bool dropping = DROP;

    const int T = a.T, Tp = a.Tp, H = a.H;
    const int ktiles = (T + KB - 1) / KB;
    const int total = a.n_items * H * ktiles;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int kblk = lin % ktiles;
    const int nh = lin / ktiles;
    const int n = nh / H, h = nh % H;
    const int mb = n % a.mask_mod;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int key = kblk * KB + wave * 32 + l31;
    const bool kok = key < T;
    const int kvend = a.kv_end[mb];
    if (kblk * KB >= kvend) {       // ragged batch: every key of this block is past the item's last valid frame (P = 0): dk = dv = 0
        if (kok) {
            float* d1 = a.dk + ((size_t)nh * T + key) * 64 + hi * 32;
            float* d2 = a.dv + ((size_t)nh * T + key) * 64 + hi * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) { *(float4*)(d1 + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f); *(float4*)(d2 + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        return;
    }

    const unsigned char* qbase = (const unsigned char*)a.q + ((size_t)nh * T) * 128;
    const unsigned char* kbase = (const unsigned char*)a.k + ((size_t)nh * T) * 128;
    const unsigned char* vbase = (const unsigned char*)a.v + ((size_t)nh * T) * 128;
    const unsigned char* vlbase = (const unsigned char*)a.vlo + ((size_t)nh * T) * 128;
    const unsigned char* qtbase = (const unsigned char*)a.qT + ((size_t)nh * 64) * Tp * 2;
    const unsigned char* dotbase = (const unsigned char*)a.dOT + ((size_t)nh * 64) * Tp * 2;
    const unsigned char* dobase = (const unsigned char*)a.dO + ((size_t)n * T) * a.dO_row_stride * 2 + (size_t)h * 128;
    const unsigned char* zeros = (const unsigned char*)a.zeros;

    vec8 kf[4], vf[4], vlf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0), wl = make_uint4(0, 0, 0, 0);
        if (kok) {
            v = *(const uint4*)(kbase + (size_t)key * 128 + ks * 32 + hi * 16);
            w = *(const uint4*)(vbase + (size_t)key * 128 + ks * 32 + hi * 16);
            wl = *(const uint4*)(vlbase + (size_t)key * 128 + ks * 32 + hi * 16);
        }
        kf[ks] = as_vec8<P>(v); vf[ks] = as_vec8<P>(w); vlf[ks] = as_vec8<P>(wl);
    }
    float alpha = 1.f;      // power-of-two operand scale of dS for this (item, head), from the dQ kernel's bound
    {
        const float bound = __uint_as_float(a.dsmax[nh]);
        if (bound > 0.f) {
            int ex = 10 - (int)ceilf(log2f(bound));
            ex = ex < -40 ? -40 : (ex > 80 ? 80 : ex);
            alpha = exp2f((float)ex);
        }
    }
    const float bias_k = kok ? a.kbias[(size_t)mb * Tp + key] : -1e30f;
    // the key's bias (0 / masked) rides in the augmented k-step of the S MFMA, next to -lse of the query rows (aug_neg3): p = exp2(acc)
    const vec8 kaug = aug_ones_plus<P>(fmaxf(bias_k, P::kMaskedScore), hi);
    const unsigned drop_ch = DROP ? a.drop.colh[(key < Tp ? key : Tp - 1) >> 1] : 0u;
    const unsigned drop_shl = (key & 1) ? 0u : 16u, drop_thr = a.drop.thresh16 << 16;      // (odd key: high half of the pair hash)

    const int nq = ((kvend < T ? kvend : T) + 63) >> 6;      // queries past the last valid frame have d attn = 0: they add nothing
    auto issue = [&](int qt, int buf) {     // 4 tiles x 8 pieces over 4 waves: each wave moves pieces 2*wave, 2*wave+1 of every tile
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wave * 2 + k;
            dma_rows(qbase, 128, qt * 64, T, zeros, Qs + buf * kTile, piece, lane);
            dma_rows(dobase, (size_t)a.dO_row_stride * 2, qt * 64, T, zeros, dOs + buf * kTile, piece, lane);
            dma_cols(qtbase, Tp, qt * 64, QTs + buf * kTile, piece, lane);
            dma_cols(dotbase, Tp, qt * 64, dOTs + buf * kTile, piece, lane);
        }
        if (wave == 0) {
            const int qq = qt * 64 + lane;
            // per-query constants, folded with this (item, head)'s operand scale: D_t = (D' + a F) alpha (no dropout: D' alpha), a_t = a alpha
            const float Dv = qq < T ? a.Dq[(size_t)nh * T + qq] : 0.f, Fv = qq < T ? a.Fq[(size_t)nh * T + qq] : 1.f, av_ = qq < T ? a.aq[(size_t)nh * T + qq] : 0.f;
            lse_t[buf * 64 + lane] = qq < T ? a.lse[(size_t)nh * T + qq] : 0.f;
            D_t[buf * 64 + lane] = (dropping ? Dv + av_ * Fv : Dv) * alpha;
            a_t[buf * 64 + lane] = av_ * alpha;
            if (DROP) rh_t[buf * 64 + lane] = a.drop.rowh[(size_t)nh * T + (qq < T ? qq : T - 1)];
        }
    };
    int row_off[2], swz[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) { const int row = b * 32 + l31; row_off[b] = row * 128; swz[b] = (row >> 1) & 7; }

    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
    float cs = 0.f;          // sum_q dS[q][key] (fp32)

    issue(0, 0);
    ST_DMA_WAIT(0);
    __syncthreads();
    for (int qt = 0; qt < nq; ++qt) {
        const int buf = qt & 1;
        if (qt + 1 < nq) issue(qt + 1, buf ^ 1);
        // One 32-query half of the tile at a time, start to finish (round 6): S and dP of the half, its element-wise step, and the dV / dK
        // MFMAs whose k-slots are those 32 queries.  Only one half's S / dP accumulators and P / dS operands are live (48 registers fewer
        // than with both halves in flight: 250 -> ~200 VGPRs), which is what leaves room to issue a half's fragment reads as batches in
        // front of their MFMAs instead of hipcc's read -> wait -> MFMA chain.  Every accumulator still sees its MFMAs in the same order
        // (dv[d] / dk[d]: g = 0, 1, 2, 3): bit-identical to the two-halves-at-once form.
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            {
                const vec8 qaug = aug_neg3<P>(lse_t[buf * 64 + qb * 32 + l31], hi);      // row = query qb * 32 + l31 of the tile
                s = P::mfma(qaug, kaug, s);
                vec8 fq[4], fo[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    fq[ks] = frag<P>(Qs + buf * kTile, row_off[qb], swz[qb], ks * 2 + hi);
                    fo[ks] = frag<P>(dOs + buf * kTile, row_off[qb], swz[qb], ks * 2 + hi);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = P::mfma(fq[ks], kf[ks], s);
                    dp = P::mfma(fo[ks], vf[ks], dp);
                    if constexpr (!kNoVlo) dp = P::mfma(fo[ks], vlf[ks], dp);
                }
            }
            vec8 pdf[2], dsf[2];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 dz = *(const float4*)(D_t + buf * 64 + qb * 32 + 8 * g4 + 4 * hi);
                const float4 az = *(const float4*)(a_t + buf * 64 + qb * 32 + 8 * g4 + 4 * hi);
                const float dvv[4] = {dz.x, dz.y, dz.z, dz.w};
                const float av[4] = {az.x, az.y, az.z, az.w};
                float fq4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (DROP) {     // this lane's key is one half of its pair: its own 16 bits of every query's pair hash (DropCfg, launch.h)
                    const uint4 rh = *(const uint4*)(rh_t + buf * 64 + qb * 32 + 8 * g4 + 4 * hi);
                    const unsigned rhv[4] = {rh.x, rh.y, rh.z, rh.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)      // the lane's half moved to the top 16 bits: one compare against thresh16 << 16
                        fq4[e] = (drop_pair(rhv[e], drop_ch) << drop_shl) >= drop_thr ? a.drop.scale : 0.0f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const float p = __builtin_amdgcn_exp2f(s[r]);      // s = q.k + bias_k - lse_q straight from the MFMAs
                    const float f = fq4[e];
                    const float pf = p * f;
                    pdf[r >> 3][r & 7] = to16<P>(pf);
                    // alpha dS = p f (dP' + a) alpha - p (D' + a F) alpha: the per-query constants arrive folded (issue()); cs accumulates SCALED values
                    float ds;
                    if (dropping) ds = pf * fmaf(dp[r], alpha, av[e]) - p * dvv[e];
                    else ds = p * fmaf(dp[r], alpha, -dvv[e]);
                    cs += ds;
                    dsf[r >> 3][r & 7] = to16<P>(ds);
                }
            }
            {   // dV^T / dK^T += (dO^T | Q^T)[d][the half's queries] . (P | dS): k-slot groups g = 2 qb, 2 qb + 1
                vec8 fa[2][2], fb[2][2];
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        fa[d][g2] = frag<P>(dOTs + buf * kTile, row_off[d], swz[d], (qb * 2 + g2) * 2 + hi);
                        fb[d][g2] = frag<P>(QTs + buf * kTile, row_off[d], swz[d], (qb * 2 + g2) * 2 + hi);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        dv[d] = P::mfma(fa[d][g2], pdf[g2], dv[d]);
                        dk[d] = P::mfma(fb[d][g2], dsf[g2], dk[d]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);      // (keeps the second half's S / dP from being hoisted above: that would restore the register pressure)
        }
        ST_DMA_WAIT(0);
        __syncthreads();
    }
    {   // undo the operand scale; Q^T was centred: add qmean[d] * sum_q dS back (fp32)
        const float inv_alpha = 1.0f / alpha;
        cs = xor32_sum(cs) * inv_alpha;        // (accumulated in scaled units)
        const float* qm = a.qmean + (size_t)nh * 64;
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2)
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[d2][r] = dk[d2][r] * inv_alpha + qm[d2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] * cs;
    }
    if (kok) {
        store_acc_rows(a.dk + ((size_t)nh * T + key) * 64, dk, hi);
        store_acc_rows(a.dv + ((size_t)nh * T + key) * 64, dv, hi);
    }
    if (a.gmax) { publish_absmax(dk, kok, a.gmax + 1); publish_absmax(dv, kok, a.gmax + 2); }
}

hipError_t launch_attn_bwd_dq(int dtype, const AttnBwdArgs& a, hipStream_t s) {
    if (!a.zeros || !a.kbias || !a.lse || !a.Dq || !a.Fq || !a.aq || !a.vmean || !a.kmean || !a.vlo || !a.dsmax) return hipErrorInvalidValue;
    // a.dsmax ([n_items * H] maxima the first pass publishes) must arrive ZEROED: the engine zeroes the cells of all blocks in one memset
    const int qtiles = (a.T + 255) / 256;
    const int total = a.n_items * a.H * qtiles;
    const int grid = ((total + 7) / 8) * 8;
    const bool drop = a.drop.thresh16 != 0;
    if (drop && (!a.drop.rowh || !a.drop.colh)) return hipErrorInvalidValue;
    if (!a.alphaq) return hipErrorInvalidValue;
    if (dtype == DT_BF16) {
        if (drop) { hipLaunchKernelGGL((attn_bwd_dq_kernel<OpBF16, true, 1>), dim3(grid), dim3(512), 0, s, a);
                    hipLaunchKernelGGL((attn_bwd_dq_kernel<OpBF16, true, 2>), dim3(grid), dim3(512), 0, s, a); }
        else      { hipLaunchKernelGGL((attn_bwd_dq_kernel<OpBF16, false, 1>), dim3(grid), dim3(512), 0, s, a);
                    hipLaunchKernelGGL((attn_bwd_dq_kernel<OpBF16, false, 2>), dim3(grid), dim3(512), 0, s, a); }
    } else {
        if (drop) { hipLaunchKernelGGL((attn_bwd_dq_kernel<OpF16, true, 1>), dim3(grid), dim3(512), 0, s, a);
                    hipLaunchKernelGGL((attn_bwd_dq_kernel<OpF16, true, 2>), dim3(grid), dim3(512), 0, s, a); }
        else      { hipLaunchKernelGGL((attn_bwd_dq_kernel<OpF16, false, 1>), dim3(grid), dim3(512), 0, s, a);
                    hipLaunchKernelGGL((attn_bwd_dq_kernel<OpF16, false, 2>), dim3(grid), dim3(512), 0, s, a); }
    }
    return hipGetLastError();
}

hipError_t launch_attn_bwd_dkv(int dtype, const AttnBwdArgs& a, hipStream_t s) {
    if (!a.zeros || !a.kbias || !a.lse || !a.Dq || !a.Fq || !a.aq || !a.qmean || !a.vlo || !a.dsmax) return hipErrorInvalidValue;
    const int ktiles = (a.T + 127) / 128;
    const int total = a.n_items * a.H * ktiles;
    const int grid = ((total + 7) / 8) * 8;
    // 66 KB of dynamic LDS: above the 64 KB default, the opt-in is per device and per kernel
    static bool attr_done_dev[64][4] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    const bool drop = a.drop.thresh16 != 0;
    if (drop && (!a.drop.rowh || !a.drop.colh)) return hipErrorInvalidValue;
    const int di = (dtype == DT_BF16 ? 0 : 1) * 2 + (drop ? 1 : 0);
    const void* fns[4] = {(const void*)attn_bwd_dkv_kernel<OpBF16, false>, (const void*)attn_bwd_dkv_kernel<OpBF16, true>,
                          (const void*)attn_bwd_dkv_kernel<OpF16, false>, (const void*)attn_bwd_dkv_kernel<OpF16, true>};
    if (!attr_done_dev[dev_][di]) {
        hipError_t e = hipFuncSetAttribute(fns[di], hipFuncAttributeMaxDynamicSharedMemorySize, kDkvLds);
        if (e != hipSuccess) return e;
        attr_done_dev[dev_][di] = true;
    }
    if (di == 0)      hipLaunchKernelGGL((attn_bwd_dkv_kernel<OpBF16, false>), dim3(grid), dim3(256), kDkvLds, s, a);
    else if (di == 1) hipLaunchKernelGGL((attn_bwd_dkv_kernel<OpBF16, true>), dim3(grid), dim3(256), kDkvLds, s, a);
    else if (di == 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<OpF16, false>), dim3(grid), dim3(256), kDkvLds, s, a);
    else              hipLaunchKernelGGL((attn_bwd_dkv_kernel<OpF16, true>), dim3(grid), dim3(256), kDkvLds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ operand copies
// natural rows -> T layout.  Source rows may be strided (time-major dO: row stride H*64, head offset h*64).
// mean[nh][d] = mean over t < T of nat[t][d]  (natural [item][H][T][64])
// one block per (item, head): thread (row group ty of 32, lane tx of 8) reads 16 bytes = head dims tx*8 .. +8 of row t,
// four rows in flight; the 32 row-group partials are combined in a fixed order (deterministic)
template <class P>
__device__ __forceinline__ void attn_mean_nat_body(const typename P::elem* nat, int T, float* mean, int nh) {
    __shared__ float part[32][64 + 1];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const typename P::elem* base = nat + (size_t)nh * T * 64 + tx * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int t = ty;
    for (; t + 96 < T; t += 128) {
        uint4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *(const uint4*)(base + (size_t)(t + 32 * u) * 64);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const typename P::vec8 x = as_vec8<P>(r[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)x[e];
        }
    }
    for (; t < T; t += 32) {
        const typename P::vec8 x = as_vec8<P>(*(const uint4*)(base + (size_t)t * 64));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)x[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[ty][tx * 8 + e] = v[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) s += part[g][threadIdx.x];
        mean[(size_t)nh * 64 + threadIdx.x] = s / (float)T;
    }
}
template <class P>
__global__ __launch_bounds__(256) void attn_mean_nat_kernel(const typename P::elem* nat, int T, float* mean) {
    attn_mean_nat_body<P>(nat, T, mean, blockIdx.x);
}

hipError_t launch_attn_mean_nat(int dtype, const void* nat, int n_heads_total, int T, float* mean, hipStream_t s) {
    if (dtype == DT_BF16) hipLaunchKernelGGL((attn_mean_nat_kernel<OpBF16>), dim3(n_heads_total), dim3(256), 0, s, (const __bf16*)nat, T, mean);
    else                  hipLaunchKernelGGL((attn_mean_nat_kernel<OpF16>), dim3(n_heads_total), dim3(256), 0, s, (const _Float16*)nat, T, mean);
    return hipGetLastError();
}

template <class P>
__device__ __forceinline__ void attn_to_T_body(const typename P::elem* nat, int64_t item_stride, int64_t head_stride,
                                               int row_stride, int H, int T, int Tp, const float* mean,
                                               typename P::elem* outT, int bx, int nh) {
    __shared__ typename P::elem tile[64][64 + 2];
    const int p0 = bx * 64;
    const int n = nh / H, h = nh % H;
    const typename P::elem* src = nat + (size_t)n * item_stride + (size_t)h * head_stride;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int pos = p0 + rr;
        tile[rr][tx] = pos < T ? src[(size_t)pos * row_stride + tx] : (typename P::elem)0.0f;
    }
    __syncthreads();
    // column c of the output holds position perm(c): bits 2 <-> 3 swapped inside every 16
    const int c = tx;
    const int pc = (c & ~12) | ((c & 4) << 1) | ((c & 8) >> 1);
    for (int d = ty; d < 64; d += 4) {
        float v = (float)tile[pc][d];
        if (mean && p0 + pc < T) v -= mean[(size_t)nh * 64 + d];        // centred copy (zero tail stays zero)
        outT[((size_t)nh * 64 + d) * Tp + p0 + c] = to16<P>(v);
    }
}
template <class P>
__global__ __launch_bounds__(256) void attn_to_T_kernel(const typename P::elem* nat, int64_t item_stride, int64_t head_stride,
                                                        int row_stride, int H, int T, int Tp, const float* mean,
                                                        typename P::elem* outT) {
    attn_to_T_body<P>(nat, item_stride, head_stride, row_stride, H, T, Tp, mean, outT, blockIdx.x, blockIdx.y);
}

hipError_t launch_attn_to_T(int dtype, const void* nat, int64_t item_stride, int64_t head_stride, int row_stride,
                            int n_items, int H, int T, int Tp, const float* mean, void* outT, hipStream_t s) {
    dim3 grid(Tp / 64, n_items * H);
    if (dtype == DT_BF16) hipLaunchKernelGGL((attn_to_T_kernel<OpBF16>), grid, dim3(256), 0, s, (const __bf16*)nat, item_stride, head_stride, row_stride, H, T, Tp, mean, (__bf16*)outT);
    else                  hipLaunchKernelGGL((attn_to_T_kernel<OpF16>), grid, dim3(256), 0, s, (const _Float16*)nat, item_stride, head_stride, row_stride, H, T, Tp, mean, (_Float16*)outT);
    return hipGetLastError();
}

// c[nh][d] = mean over the positions t < T of V[t][d]  (V^T in the T layout: the zero tail adds nothing)
template <class P, bool LO = false>
__device__ __forceinline__ void attn_vmean_body(const typename P::elem* inT, const typename P::elem* inT_lo, int T, int Tp, float* vmean, int nh) {
    // one block per (item, head); wave w handles head dims w, w+4, ...; lanes stride over the positions
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = wave; d < 64; d += 4) {          // 16 bytes (8 positions) per lane and load; Tp is a multiple of 64
        const typename P::elem* row = inT + ((size_t)nh * 64 + d) * Tp;
        float v = 0.f;
        for (int c = lane * 8; c < Tp; c += 512) {
            const typename P::vec8 x = as_vec8<P>(*(const uint4*)(row + c));
#pragma unroll
            for (int e = 0; e < 8; ++e) v += (float)x[e];
            if constexpr (LO) {       // v = hi + lo (training forward with hi + lo v operands)
                const typename P::vec8 y = as_vec8<P>(*(const uint4*)(inT_lo + ((size_t)nh * 64 + d) * Tp + c));
#pragma unroll
                for (int e = 0; e < 8; ++e) v += (float)y[e];
            }
        }
        v = wave_sum(v);
        if (lane == 0) vmean[(size_t)nh * 64 + d] = v / (float)T;
    }
}
template <class P>
__global__ __launch_bounds__(256) void attn_vmean_kernel(const typename P::elem* inT, int T, int Tp, float* vmean) {
    attn_vmean_body<P>(inT, nullptr, T, Tp, vmean, blockIdx.x);
}

// T layout -> natural rows, centred: nat[t][d] = V^T[d][perm(t)] - c[d]
template <class P, bool LO = false>
__device__ __forceinline__ void attn_from_T_body(const typename P::elem* inT, const typename P::elem* inT_lo, int T, int Tp, const float* vmean,
                                                 typename P::elem* nat, typename P::elem* nat_lo, int bx, int nh) {
    __shared__ typename P::elem tile[64][64 + 2];
    __shared__ typename P::elem tile_lo[LO ? 64 : 1][64 + 2];
    const int p0 = bx * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int d = ty; d < 64; d += 4) {      // column tx = position perm(tx)
        tile[d][tx] = inT[((size_t)nh * 64 + d) * Tp + p0 + tx];
        if constexpr (LO) tile_lo[d][tx] = inT_lo[((size_t)nh * 64 + d) * Tp + p0 + tx];
    }
    __syncthreads();
    const float cm = vmean ? vmean[(size_t)nh * 64 + tx] : 0.f;
    for (int rr = ty; rr < 64; rr += 4) {
        const int pos = p0 + rr;
        const int c = (rr & ~12) | ((rr & 4) << 1) | ((rr & 8) >> 1);     // perm is an involution: position rr sits in column perm(rr)
        if (pos < T) {
            float v = (float)tile[tx][c] - cm;
            if constexpr (LO) v += (float)tile_lo[tx][c];
            const typename P::elem hi16 = to16<P>(v);
            nat[((size_t)nh * T + pos) * 64 + tx] = hi16;
            if (nat_lo) nat_lo[((size_t)nh * T + pos) * 64 + tx] = to16<P>(v - (float)hi16);
        }
    }
}
template <class P>
__global__ __launch_bounds__(256) void attn_from_T_kernel(const typename P::elem* inT, int T, int Tp, const float* vmean,
                                                          typename P::elem* nat, typename P::elem* nat_lo) {
    attn_from_T_body<P>(inT, nullptr, T, Tp, vmean, nat, nat_lo, blockIdx.x, blockIdx.y);
}

// The gradient-independent operand copies of one attention backward in TWO launches (were six): the three means (grid.y = q, k, v),
// then the centred copies (grid.z = V back to natural rows as a hi + lo pair, Q^T, K^T).  Same bodies, same results.
template <class P, bool VLO>
__global__ __launch_bounds__(256) void attn_prep_means_kernel(const typename P::elem* q, const typename P::elem* k, const typename P::elem* vT,
                                                              const typename P::elem* vT_lo, int T, int Tp, float* qmean, float* kmean, float* vmean) {
    if (blockIdx.y == 0) attn_mean_nat_body<P>(q, T, qmean, blockIdx.x);
    else if (blockIdx.y == 1) attn_mean_nat_body<P>(k, T, kmean, blockIdx.x);
    else attn_vmean_body<P, VLO>(vT, vT_lo, T, Tp, vmean, blockIdx.x);
}
template <class P, bool VLO>
__global__ __launch_bounds__(256) void attn_prep_copies_kernel(const typename P::elem* q, const typename P::elem* k, const typename P::elem* vT,
                                                               const typename P::elem* vT_lo, int H, int T, int Tp, const float* qmean, const float* kmean, const float* vmean,
                                                               typename P::elem* qT, typename P::elem* kT, typename P::elem* vnat, typename P::elem* vnat_lo) {
    if (blockIdx.z == 0) attn_from_T_body<P, VLO>(vT, vT_lo, T, Tp, vmean, vnat, vnat_lo, blockIdx.x, blockIdx.y);
    else if (blockIdx.z == 1) attn_to_T_body<P>(q, (int64_t)H * T * 64, (int64_t)T * 64, 64, H, T, Tp, qmean, qT, blockIdx.x, blockIdx.y);
    else attn_to_T_body<P>(k, (int64_t)H * T * 64, (int64_t)T * 64, 64, H, T, Tp, kmean, kT, blockIdx.x, blockIdx.y);
}
namespace {
template <class P, bool VLO>
void attn_prep_launch(const void* q, const void* k, const void* vT, const void* vT_lo, int n_items, int H, int T, int Tp, float* qmean, float* kmean,
                      float* vmean, void* qT, void* kT, void* vnat, void* vnat_lo, hipStream_t s) {
    using E = typename P::elem;
    const dim3 g1(n_items * H, 3), g2(Tp / 64, n_items * H, 3);
    hipLaunchKernelGGL((attn_prep_means_kernel<P, VLO>), g1, dim3(256), 0, s, (const E*)q, (const E*)k, (const E*)vT, (const E*)vT_lo, T, Tp, qmean, kmean, vmean);
    hipLaunchKernelGGL((attn_prep_copies_kernel<P, VLO>), g2, dim3(256), 0, s, (const E*)q, (const E*)k, (const E*)vT, (const E*)vT_lo, H, T, Tp, qmean, kmean, vmean,
                       (E*)qT, (E*)kT, (E*)vnat, (E*)vnat_lo);
}
}  // namespace
hipError_t launch_attn_prep(int dtype, const void* q, const void* k, const void* vT, const void* vT_lo, int n_items, int H, int T, int Tp, float* qmean,
                            float* kmean, float* vmean, void* qT, void* kT, void* vnat, void* vnat_lo, hipStream_t s) {
    if (dtype == DT_BF16) {
        if (vT_lo) attn_prep_launch<OpBF16, true>(q, k, vT, vT_lo, n_items, H, T, Tp, qmean, kmean, vmean, qT, kT, vnat, vnat_lo, s);
        else       attn_prep_launch<OpBF16, false>(q, k, vT, vT_lo, n_items, H, T, Tp, qmean, kmean, vmean, qT, kT, vnat, vnat_lo, s);
    } else {
        if (vT_lo) attn_prep_launch<OpF16, true>(q, k, vT, vT_lo, n_items, H, T, Tp, qmean, kmean, vmean, qT, kT, vnat, vnat_lo, s);
        else       attn_prep_launch<OpF16, false>(q, k, vT, vT_lo, n_items, H, T, Tp, qmean, kmean, vmean, qT, kT, vnat, vnat_lo, s);
    }
    return hipGetLastError();
}

hipError_t launch_attn_from_T(int dtype, const void* inT, int n_items, int H, int T, int Tp, float* vmean, void* nat, void* nat_lo,
                              hipStream_t s) {
    dim3 grid(Tp / 64, n_items * H);
    if (dtype == DT_BF16) {
        if (vmean) hipLaunchKernelGGL((attn_vmean_kernel<OpBF16>), dim3(n_items * H), dim3(256), 0, s, (const __bf16*)inT, T, Tp, vmean);
        hipLaunchKernelGGL((attn_from_T_kernel<OpBF16>), grid, dim3(256), 0, s, (const __bf16*)inT, T, Tp, vmean, (__bf16*)nat, (__bf16*)nat_lo);
    } else {
        if (vmean) hipLaunchKernelGGL((attn_vmean_kernel<OpF16>), dim3(n_items * H), dim3(256), 0, s, (const _Float16*)inT, T, Tp, vmean);
        hipLaunchKernelGGL((attn_from_T_kernel<OpF16>), grid, dim3(256), 0, s, (const _Float16*)inT, T, Tp, vmean, (_Float16*)nat, (_Float16*)nat_lo);
    }
    return hipGetLastError();
}

// dq_acc, dk_acc, dv (natural fp32 [item][H][T][64]) -> time-major 16-bit [item][T][3*H*64] = [dq | dk | dv] of the fused
// QKV projection: d q_proj = R(-theta)(dq_acc / 8), d k_proj = R(-theta)(dk_acc ln2), d v_proj = dv, where R(theta) is the
// partial rotary embedding of the forward (pairs (j, j+16), j < 16; models/diffusion_transformer.py:180-198).
template <class P>
__global__ __launch_bounds__(256) void qkv_grad_pack_kernel(const float* dq, const float* dk, const float* dv,
                                                            const float* rope_cos, const float* rope_sin, int H, int T,
                                                            int64_t total, const float* qs, typename P::elem* out, typename P::elem* outw) {
    // one thread per (item, t, h, quarter c): head dims 8c .. 8c+7 and 8c+16 .. 8c+23 for c < 2 (the rotated pairs (j, j+16), j < 16),
    // dims 16c .. 16c+15 for c >= 2 (pass-through): every load is a float4, every store 16 bytes (round 6; the element-per-lane form
    // it replaces issued 2-byte stores 32 bytes apart: 90 us for 390 MB)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx & 3);
    const int h = (int)((idx >> 2) % H);
    const int64_t row = (idx >> 2) / H;               // item * T + t
    const int64_t n = row / T; const int t = (int)(row - n * T);
    const size_t src = (((size_t)n * H + h) * T + t) * 64;
    const int C = H * 64;
    const float fc = qs[0], fq = qs[8], fk = qs[9], fv = qs[10];      // powers of two: the products below are exact
    const int d0 = c < 2 ? 8 * c : 16 * c, d1 = c < 2 ? 8 * c + 16 : 16 * c + 8;      // the thread's two 8-dim chunks
    auto ld8 = [&](const float* p, float (&v)[8]) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    };
    auto st8 = [&](typename P::elem* p, const float (&v)[8], float f) {
        const uint2 lo = pack4<P>(v[0] * f, v[1] * f, v[2] * f, v[3] * f), hi = pack4<P>(v[4] * f, v[5] * f, v[6] * f, v[7] * f);
        *(uint4*)p = make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    float cs[8], sn[8];
    if (c < 2) { ld8(rope_cos + (size_t)t * 16 + 8 * c, cs); ld8(rope_sin + (size_t)t * 16 + 8 * c, sn); }
    typename P::elem* o = out + (size_t)row * 3 * C + h * 64;
    typename P::elem* w = outw + (size_t)row * 3 * C + h * 64;
    auto one = [&](const float* g, float pre, int plane, float fw, bool rot) {
        float x0[8], x1[8];
        ld8(g + src + d0, x0); ld8(g + src + d1, x1);
#pragma unroll
        for (int e = 0; e < 8; ++e) { x0[e] *= pre; x1[e] *= pre; }
        if (rot && c < 2) {      // y1 = x1 c - x2 s, y2 = x2 c + x1 s  =>  dx1 = dy1 c + dy2 s, dx2 = dy2 c - dy1 s
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a1 = x0[e], a2 = x1[e];
                x0[e] = a1 * cs[e] + a2 * sn[e]; x1[e] = a2 * cs[e] - a1 * sn[e];
            }
        }
        st8(o + plane * C + d0, x0, fc); st8(o + plane * C + d1, x1, fc);
        st8(w + plane * C + d0, x0, fw); st8(w + plane * C + d1, x1, fw);
    };
    one(dq, 0.125f, 0, fq, true);
    one(dk, 0.6931471805599453f, 1, fk, true);
    one(dv, 1.0f, 2, fv, false);
}

hipError_t launch_qkv_grad_pack(int dtype, const float* dq, const float* dk, const float* dv, const float* rope_cos,
                                const float* rope_sin, int n_items, int H, int T, const float* qs, void* d16, void* w16,
                                hipStream_t s) {
    const int64_t total = (int64_t)n_items * T * H * 4;
    const int grid = (int)((total + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((qkv_grad_pack_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, dq, dk, dv, rope_cos, rope_sin, H, T, total, qs, (__bf16*)d16, (__bf16*)w16);
    else                  hipLaunchKernelGGL((qkv_grad_pack_kernel<OpF16>), dim3(grid), dim3(256), 0, s, dq, dk, dv, rope_cos, rope_sin, H, T, total, qs, (_Float16*)d16, (_Float16*)w16);
    return hipGetLastError();
}

}  // namespace st
