// Flash-style masked self-attention for gfx950 (replaces F.scaled_dot_product_attention with the
// materialised (B,1,T,T) additive mask of models/diffusion_transformer.py:77,107-108).
//
//   q, k : [item][H][T][64]   16-bit, RoPE applied; q pre-scaled by log2(e)/sqrt(64)
//   vT   : [item][H][64][Tp]  16-bit, keys contiguous, Tp = T rounded up to 64 with a zero tail; inside
//                             every group of 16 keys, key bits 2 and 3 are swapped (see below)
//   out  : [item][T][H*64]    16-bit
//   kbias: [mask row][Tp]     fp32 additive key bias in log2 units: 0 for valid keys, -1e30 for masked
//                             or out-of-range keys (built once per solve from the (B,1,T) mask)
//
// One block = ST_ATTN_WAVES (8) waves = 256 queries of one (item, head); each wave owns 32 queries.  K / V^T tiles of
// 64 keys go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per tile;
// the LDS image is dense 128-byte rows with the source-side XOR swizzle of conv_gemm_impl.h, so every
// ds_read_b128 fragment read is bank-conflict free.
// Both MFMAs are issued "transposed" so that a QUERY IS A LANE everywhere:
//   S^T[key][query] = K . Q^T      (A = K tile from LDS, B = Q fragment held in registers)
//   O^T[d][query]  += V^T . P^T    (A = V^T tile from LDS, B = P^T = exp2(S^T - m) in registers)
// The accumulator layout of S^T (lane = query, registers = keys) IS the B-operand layout of P^T up to
// a permutation of the k-slots, and a permutation of k-slots applied to both operands does not change
// a dot product -- so P never leaves registers and never crosses lanes.  The matching key order of the
// V^T operand (bits 2<->3 of the key index swapped) is baked into the global vT layout by the QKV
// epilogue.  Softmax max/sum are lane-local plus one shuffle with lane^32 (the other half of the same
// query's keys).  The softmax reference value of a query enters the scores THROUGH THE MFMA (a fifth k-step:
// K augmented by a column of ones, Q by -m_ref), so p = exp2 of the accumulator with no per-element subtraction, and
// the reference is raised lazily (O, l rescaled) only when a tile's maximum exceeds it by 2^8: per key tile a wave
// issues 9 + 8 MFMAs next to 32 v_exp, 32 v_add (row sum), 16 v_max3 and 16 v_cvt_pk.
// Tiles below the valid prefix skip the bias add entirely, tiles past the last valid key are never
// visited.  Padded QUERY rows produce finite garbage that the out-projection epilogue multiplies by
// 0, exactly as the reference's uniform-softmax rows are zeroed by "* x_mask" (:111).
#include "common.h"
#include "launch.h"

// ablation builds of the inference kernel (tools/ab_attention_ablation.sh): the softmax without its exps / row sums / maxima
#if (defined(ST_ABL_NOEXP) || defined(ST_ABL_NOSUM)) && !defined(ST_DEVTOOLS)
#error "ST_ABL_* (ablation builds: results are garbage) need -DST_DEVTOOLS"
#endif
#ifdef ST_ABL_NOEXP
constexpr bool kAblNoExp = true;
#else
constexpr bool kAblNoExp = false;
#endif
#ifdef ST_ABL_NOSUM
constexpr bool kAblNoSum = true;
#else
constexpr bool kAblNoSum = false;
#endif
#ifndef ST_ATTN_WAVES
#define ST_ATTN_WAVES 8      // waves (x 32 queries) per block sharing one K/V tile stream: 4, 8 or 16
#endif

namespace st {

// TRAIN: also writes the log2-sum-exp of every query row (for the backward's recomputation of P) and applies
// dropout to the probabilities that enter P.V (not to the normaliser), as SDPA's dropout_p does.
// SPLIT (inference, `attention_precision = split`): q and k arrive as hi + lo pairs of 16-bit operands and the scores are formed as
// q_hi k_hi + q_lo k_hi + q_hi k_lo (fp32 accumulation; the lo x lo term is below fp32's own rounding of the sum): 3x the QK^T MFMAs,
// one more K tile per stage in LDS, 16 more registers for the q_lo fragments -- for checkpoints whose softmax is an arg-max
// (score maxima of 80-200), where the 2^-11 rounding of q and k moves the winning probability.
// VLO (training): v arrives as a hi + lo pair (vt, vt_lo) and the output is P v_hi + P v_lo -- the rounding of v is the one forward operand
// rounding the conv_q / conv_k weight gradients are ill-conditioned in at random init (a key-independent part of v cancels in dP - D, its
// rounding error does not; tools/train_qk_split_estimate.py): 2x the PV MFMAs, one more V^T tile per stage.
template <class P, bool TRAIN, bool SPLIT = false, bool VLO = false>
__global__ __launch_bounds__(64 * ST_ATTN_WAVES, ST_ATTN_WAVES >= 16 || SPLIT ? 2 : 4) void attention_kernel(const AttnArgs a) {
    static_assert(!(TRAIN && SPLIT), "split-precision scores: inference kernel only");
    static_assert(!VLO || (TRAIN && !SPLIT), "hi + lo v: training kernel only");
    constexpr bool X3 = SPLIT || VLO;            // a third tile per stage (K_lo or V^T_lo)
    constexpr int NW = ST_ATTN_WAVES, QB = 32 * NW, QTILE = QB;      // waves and queries per block
    using vec8 = typename P::vec8;
    constexpr int TILE_BYTES = 64 * 128;
    constexpr int NBUF = 3;                      // K / V^T tile ring: tile kt+2 is in flight while tile kt is computed
    constexpr int NT = X3 ? 3 : 2;              // tiles per stage: K, V^T (, K_lo or V^T_lo)
    constexpr int SMEM = NT * NBUF * TILE_BYTES > NW * 32 * 144 ? NT * NBUF * TILE_BYTES : NW * 32 * 144;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    __shared__ float wave_lse[NW];
    unsigned char* Ks = smem;                       // NBUF buffers
    unsigned char* Vs = smem + NBUF * TILE_BYTES;   // NBUF buffers
    unsigned char* KLs = smem + 2 * NBUF * TILE_BYTES;   // NBUF buffers (SPLIT: K_lo; VLO: V^T_lo)
    unsigned char* VLs = KLs;

    const int T = a.T, Tp = a.Tp, H = a.H;
    const int qtiles = (T + QB - 1) / QB;
    // (item, head) groups are dealt round-robin to the 8 XCDs (blockIdx & 7 = the XCD a block lands on) and a group's query tiles
    // stay on one XCD, whose L2 then holds the group's K / V once.  Round 4 gave every XCD a CONTIGUOUS range of groups: with a
    // length-sorted ragged batch (the sharder's order) one XCD got the longest utterances (work ~ len^2: up to 2.4x the shortest
    // XCD's): the class took 4.17 ms per solve instead of 3.78 on bench.py --ragged; paired in the default two-part solve the
    // round-robin deal is +0.8 % (all-ones) / +0.9 % (ragged) -- profiles/r05_ab_attn_xcd_ws_blocks.txt, r05_ab_nt_dma.txt.
    const int j = blockIdx.x >> 3;
    const int qt = j % qtiles, nh = (j / qtiles) * 8 + (blockIdx.x & 7);
    if (nh >= a.n_items * H) return;
    const int n = nh / H, h = nh % H;
    const int mb = n % a.mask_mod;
    if (a.t_lim && qt * QTILE >= a.t_lim[mb]) return;        // ragged batch: every query of this tile is past the item's last needed frame
    const int kvend = a.kv_end[mb];
    const int nfull = a.n_full[mb];
    const float* kbias = a.kbias + (size_t)mb * Tp;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int query = qt * QB + wave * 32 + l31;

    const unsigned char* qbase = (const unsigned char*)a.q + ((size_t)nh * T) * 128;
    const unsigned char* kbase = (const unsigned char*)a.k + ((size_t)nh * T) * 128;
    const unsigned char* vbase = (const unsigned char*)a.vt + ((size_t)nh * 64) * Tp * 2;
    const unsigned char* qlbase = SPLIT ? (const unsigned char*)a.q_lo + ((size_t)nh * T) * 128 : nullptr;
    const unsigned char* klbase = SPLIT ? (const unsigned char*)a.k_lo + ((size_t)nh * T) * 128 : nullptr;
    const unsigned char* vlbase = VLO ? (const unsigned char*)a.vt_lo + ((size_t)nh * 64) * Tp * 2 : nullptr;
    if constexpr (TRAIN) {
        if (qt * QB >= kvend) {      // ragged batch: queries past the item's last valid frame -- their rows are multiplied by the
            if (query < T) {         // mask downstream (diffusion_transformer.py:111); the backward reads them: defined zeros
                unsigned char* orow = (unsigned char*)a.out + (((size_t)n * T + query) * (H * 64) + h * 64) * 2 + hi * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint4*)(orow + 16 * i) = make_uint4(0, 0, 0, 0);
                if (hi == 0) a.lse[(size_t)nh * T + query] = 0.f;
            }
            return;
        }
    }

    // Q fragments (B operand): lane (query, hi) holds head dims ks*16 + hi*8 .. +8
    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (query < T) v = *(const uint4*)(qbase + (size_t)query * 128 + ks * 32 + hi * 16);
        qf[ks] = as_vec8<P>(v);
    }
    vec8 qlf[SPLIT ? 4 : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (query < T) v = *(const uint4*)(qlbase + (size_t)query * 128 + ks * 32 + hi * 16);
            qlf[ks] = as_vec8<P>(v);
        }
    }

    const int ntiles = (kvend + 63) >> 6;
    // LDS-DMA: the 8 + 8 pieces (8 rows x 128 B each) of the K tile and of the V^T tile are split over the waves
    // (SGPR base + 32-bit per-lane offset: the per-lane parts are loop invariant single registers; out-of-range K rows of
    //  the last tile are clamped to key T-1 -- a valid row whose score the key bias sends to -1e30 anyway)
    auto sgpr_ptr = [](const unsigned char* p) {
        const uintptr_t v = (uintptr_t)p;
        return (const unsigned char*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
    };
    auto issueKV = [&](int kt, int buf) {
        auto k_piece = [&](int piece) {
            int ln = lane;
            if constexpr (TRAIN) asm volatile("" : "+v"(ln));      // (recomputed per tile instead of held: the training variant has no spare register)
            const int row = piece * 8 + (ln >> 3);
            const int seg = (ln & 7) ^ ((row >> 1) & 7);
            const int key = min(kt * 64 + row, T - 1);
            glds16s(sgpr_ptr(kbase), (unsigned)(key * 128 + seg * 16), Ks + buf * TILE_BYTES + piece * 1024);
            if constexpr (SPLIT) glds16s(sgpr_ptr(klbase), (unsigned)(key * 128 + seg * 16), KLs + buf * TILE_BYTES + piece * 1024);
        };
        auto v_piece = [&](int piece) {
            // V^T row = head dim `row`; 8 consecutive (permuted) keys kt*64 + seg*8 .. +8, always inside Tp
            const int row = piece * 8 + (lane >> 3);
            const int seg = (lane & 7) ^ ((row >> 1) & 7);
            glds16s(sgpr_ptr(vbase + (size_t)kt * 128), (unsigned)((row * Tp + seg * 8) * 2), Vs + buf * TILE_BYTES + piece * 1024);
            if constexpr (VLO) glds16s(sgpr_ptr(vlbase + (size_t)kt * 128), (unsigned)((row * Tp + seg * 8) * 2), VLs + buf * TILE_BYTES + piece * 1024);
        };
        if constexpr (NW <= 8) {
#pragma unroll
            for (int k = 0; k < 8 / NW; ++k) { k_piece(wave * (8 / NW) + k); v_piece(wave * (8 / NW) + k); }
        } else {        // 16 waves: waves 0..7 move the K pieces, waves 8..15 the V^T pieces
            if (wave < 8) k_piece(wave); else v_piece(wave - 8);
        }
    };

    // fragment read addressing (row = l31 + 32*block): byte = row*128 + (seg ^ ((row>>1)&7))*16
    int row_off[2], swz[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int row = b * 32 + l31;
        row_off[b] = row * 128; swz[b] = (row >> 1) & 7;
    }

    f32x16_t o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    // Online softmax against a per-query REFERENCE m_ref (a 16-bit representable value, raised lazily) that enters the
    // scores through the MFMA instead of the VALU: the K tile is augmented by one k-step whose slot 0 is 1 for every key
    // and whose slot 1 is 1 for masked keys, Q by (-m_ref, -BIG) -- so the QK^T accumulator comes out as
    // s' = s - m_ref + key bias and p = exp2(s') needs no subtraction and no bias add (a per-row constant cancels
    // exactly between P and l, so the 16-bit rounding of m_ref is harmless; the product 1 x (-m_ref) is exact).
    // m_ref is raised (O, l rescaled, s' recomputed) only when a lane's partial row sum of a tile exceeds kBig = 2^13: every
    // p then fits f16 (and bf16) with room to spare, and since m_ref >= the first tile's exact maximum, smaller terms only
    // underflow when they are negligible (2^-27 of the largest).  The first tile always takes the correction path (m_ref
    // starts at 0).
    constexpr float kBig = 8192.0f, kLazyTrain = 8.0f, kFloor = -20000.0f;
    unsigned drop_rh = 0;
    if constexpr (TRAIN) { if (a.drop.thresh16) drop_rh = a.drop.rowh[(size_t)nh * T + (query < T ? query : T - 1)]; }
    float m_ref = 0.f, l_run = 0.f;
    vec8 qaug, kaug;
#pragma unroll
    for (int i = 0; i < 8; ++i) { qaug[i] = to16<P>(0.f); kaug[i] = to16<P>(0.f); }
    if (hi == 0) kaug[0] = to16<P>(1.0f);

    // Counted waits: every wave issues the same number of 1-KiB pieces per tile, so `vmcnt(pieces of one tile)` retires
    // everything but the youngest tile.  (With two buffers and a full drain per tile the iteration time was the LDS-DMA
    // round trip, not the tile's MFMA + softmax work.)
    if (ntiles > 0) issueKV(0, 0);
    if (ntiles > 1) { issueKV(1, 1); if constexpr (NW <= 8) { if constexpr (NW == 8) { if constexpr (X3) ST_DMA_WAIT(3); else ST_DMA_WAIT(2); } else { if constexpr (X3) ST_DMA_WAIT(6); else ST_DMA_WAIT(4); } } else ST_DMA_WAIT(1); }
    else ST_DMA_WAIT(0);
    __syncthreads();

    const int ntiles_run = ntiles;
    // ragged batch, inference: a wave whose 32 queries all lie past the item's last needed frame only keeps moving its share of
    // the K / V pieces and taking the barriers -- its SIMD's matrix and vector issue slots go to the waves with real queries
    bool dead = false;
    if constexpr (!TRAIN) dead = a.t_lim && qt * QB + wave * 32 >= a.t_lim[mb];
    int buf = 0;
    for (int kt = 0; kt < ntiles_run; ++kt) {
        if (kt + 2 < ntiles_run) issueKV(kt + 2, buf >= 1 ? buf - 1 : NBUF - 1);      // (buf + 2) % 3
        if (!dead) {

        f32x16_t s[2];
        const bool partial = (kt + 1) * 64 > nfull;       // tiles inside the valid prefix have no masked key
        // ---- S'^T = [K | 1] . [Q | -m_ref]^T (+ key bias)  (log2 units)
        auto scores = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
                s[kb] = P::mfma(kaug, qaug, s[kb]);
                const unsigned char* kp = Ks + buf * TILE_BYTES + row_off[kb];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const vec8 kfrag = as_vec8<P>(*(const uint4*)(kp + (((ks * 2 + hi) ^ swz[kb]) << 4)));
                    s[kb] = P::mfma(kfrag, qf[ks], s[kb]);
                    if constexpr (SPLIT) {      // + q_lo k_hi + q_hi k_lo
                        s[kb] = P::mfma(kfrag, qlf[ks], s[kb]);
                        const unsigned char* klp = KLs + buf * TILE_BYTES + row_off[kb];
                        s[kb] = P::mfma(as_vec8<P>(*(const uint4*)(klp + (((ks * 2 + hi) ^ swz[kb]) << 4))), qf[ks], s[kb]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (partial) {        // key bias (-1e30 on masked / out-of-range keys): only tiles that reach past the valid prefix
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 bz = *(const float4*)(kbias + kt * 64 + kb * 32 + 8 * g4 + 4 * hi);
                        s[kb][4 * g4 + 0] += bz.x; s[kb][4 * g4 + 1] += bz.y;
                        s[kb][4 * g4 + 2] += bz.z; s[kb][4 * g4 + 3] += bz.w;
                    }
            }
        };
        // ---- softmax numerators against the lazily raised reference
        vec8 pf[4];
        float psum[2];
        scores();
        if constexpr (TRAIN) {
            // Training variant (128 VGPRs with the dropout state; kept as straight-line code -- the lambda form of the
            // inference branch below costs it two spilled registers): the per-tile maximum and its check sit in front of the exps.
            float mx = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = xor32_max(mx);
            if (kt == 0 || __any(mx > kLazyTrain)) {
                // Rare (and the first tile): raise the reference of the rows that need it, rescale their O and l, and
                // RECOMPUTE the tile's scores with the new reference (subtracting the step from s' instead would keep the
                // fp32 rounding of s - m_ref_old).  A fully masked tile (mx = -1e30) leaves the reference at kFloor: its
                // probabilities are exactly 0 and the first tile with a valid key corrects from there.
                const bool need = kt == 0 || mx > kLazyTrain;
                const float m_new = need ? (float)to16<P>(fmaxf(m_ref + mx, kFloor)) : m_ref;
                const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(m_ref - m_new);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                m_ref = m_new;
                if (hi == 0) qaug[0] = to16<P>(-m_ref);
                scores();
            }
            psum[0] = 0.f; psum[1] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // elements 4 g4 .. 4 g4 + 3 are keys key0 .. key0 + 3 = two consecutive key pairs: one 8-byte table load, one
                    // hash per pair decides both of its elements (DropCfg, launch.h)
                    if (g4 == 2) __builtin_amdgcn_sched_barrier(0);      // (at most two table loads in flight: 128 VGPRs)
                    uint2 ch = make_uint2(0u, 0u);
                    if (a.drop.thresh16) ch = *(const uint2*)(a.drop.colh + ((kt * 64 + kb * 32 + 8 * g4 + 4 * hi) >> 1));
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int r = 4 * g4 + 2 * h2;
                        float p0 = __builtin_amdgcn_exp2f(s[kb][r]), p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
                        psum[0] = add_f32_asm_safe(psum[0], p0);
                        psum[1] = add_f32_asm_safe(psum[1], p1);
                        if (a.drop.thresh16) {
                            const float2 f = drop_factors2(a.drop, drop_pair(drop_rh, h2 ? ch.y : ch.x));
                            p0 *= f.x; p1 *= f.y;
                        }
                        pf[kb * 2 + (r >> 3)][r & 7] = to16<P>(p0);
                        pf[kb * 2 + (r >> 3)][(r & 7) + 1] = to16<P>(p1);
                    }
                }
        } else {
            // Inference: NO per-element maximum in the common path.  p = exp2(s') is formed in fp32, and only when some lane's
            // partial row sum of the tile exceeds kBig (a probability that would leave f16's comfortable range) are the maxima
            // computed, the references of the rows above theirs raised, and the tile redone.  (On a SIMD a VALU instruction is
            // issued instead of an MFMA; the 21 v_max3 / v_max per tile were 5 % of the kernel:
            // profiles/r03_attention_sq_counters_and_ablation.txt.)
            auto raise = [&](bool first) __attribute__((always_inline)) {      // same step as in the training branch, for every row above its reference
                float mx = s[0][0];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
                mx = xor32_max(mx);
                const bool need = first || mx > 0.0f;
                const float m_new = need ? (float)to16<P>(fmaxf(m_ref + mx, kFloor)) : m_ref;
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(m_ref - m_new);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                m_ref = m_new;
                if (hi == 0) qaug[0] = to16<P>(-m_ref);
                scores();
            };
            auto numerators = [&]() __attribute__((always_inline)) {
                psum[0] = 0.f; psum[1] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float p0, p1;
                        if constexpr (kAblNoExp) { p0 = s[kb][r] * 0.001f; p1 = s[kb][r + 1] * 0.001f; }
                        else { p0 = __builtin_amdgcn_exp2f(s[kb][r]); p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]); }
                        if constexpr (!kAblNoSum) {
                            psum[0] = add_f32_scalar(psum[0], p0);
                            psum[1] = add_f32_scalar(psum[1], p1);
                        }
                        pf[kb * 2 + (r >> 3)][r & 7] = to16<P>(p0);
                        pf[kb * 2 + (r >> 3)][(r & 7) + 1] = to16<P>(p1);
                    }
            };
            if (kt == 0) raise(true);        // the first tile starts from its exact maxima (m_ref starts at 0)
            numerators();
            if (kt != 0 && __any(!(psum[0] + psum[1] <= kBig))) {      // rare (the negated form also catches NaN sums: inf - inf)
                scores();           // s' was consumed by the exps: recompute it against the old reference for the maxima
                raise(false);
                numerators();
            }
        }
        l_run += psum[0] + psum[1];

        // ---- O^T += V^T . P^T
        __builtin_amdgcn_sched_barrier(0);       // (keeps the V^T fragment reads below the softmax: 32 registers)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const unsigned char* vp = Vs + buf * TILE_BYTES + row_off[d];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                o[d] = P::mfma(as_vec8<P>(*(const uint4*)(vp + (((g * 2 + hi) ^ swz[d]) << 4))), pf[g], o[d]);
            if constexpr (VLO) {
                const unsigned char* vlp = VLs + buf * TILE_BYTES + row_off[d];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    o[d] = P::mfma(as_vec8<P>(*(const uint4*)(vlp + (((g * 2 + hi) ^ swz[d]) << 4))), pf[g], o[d]);
            }
        }
        }
        // tile kt+1 (asm-issued LDS-DMA, flying under the MFMAs and exps of two tiles) has landed; tile kt+2 stays in flight
        if (kt + 2 < ntiles_run) { if constexpr (NW <= 8) { if constexpr (NW == 8) { if constexpr (X3) ST_DMA_WAIT(3); else ST_DMA_WAIT(2); } else { if constexpr (X3) ST_DMA_WAIT(6); else ST_DMA_WAIT(4); } } else ST_DMA_WAIT(1); }
        else ST_DMA_WAIT(0);
        __syncthreads();      // publishes tile kt+1, frees buffer kt % 3 for tile kt+3
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    const float m_run = m_ref;
    const float l_tot = xor32_sum(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if constexpr (!TRAIN) {
        // Run-time statistic (st_attention_stats): the largest log2-sum-exp of any VALID query row.  It bounds the row's score maximum
        // from above within log2(T) -- a serving loop reads it to see whether a checkpoint's softmax has become an arg-max (maxima of
        // 80-200 in natural units), the regime in which 16-bit q / k operands miss the 1e-3 parity bar.  One atomic per block.
        if (a.lse_max) {
            float v = -3.0e38f;
            if (!dead && query < T && query < kvend && l_tot > 0.f) v = m_run + log2f(l_tot);
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
            if (lane == 0) wave_lse[wave] = v;
            __syncthreads();
            if (tid == 0) {
                float b = wave_lse[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) b = fmaxf(b, wave_lse[w]);
                int bits = __float_as_int(b);
                bits = bits >= 0 ? bits : bits ^ 0x7fffffff;      // order-preserving map of floats onto signed ints
                atomicMax((int*)a.lse_max + ((blockIdx.x % kLseCells) << 4), bits);
            }
        }
    }
    if (dead) return;      // (after the last barrier; nothing below synchronises the block)
    if constexpr (TRAIN) {
        // (the row index is re-derived here: hipcc otherwise carries its 64-bit form across the key loop -- two registers the
        //  128-VGPR training variant does not have)
        int lane_q = lane;
        asm volatile("" : "+v"(lane_q));
        const int query_e = qt * QB + wave * 32 + (lane_q & 31);
        if ((lane_q >> 5) == 0 && query_e < T) a.lse[(size_t)nh * T + query_e] = l_tot > 0.f ? m_run + log2f(l_tot) : 0.f;
    }
    // Output through LDS: each wave parks its 32 x 64 tile as [query][d] (144-B pitch) in its own slice of the K/V
    // buffers (free after the loop's last barrier) and writes it out as 128-B rows, 16 B per lane -- 4 wide stores
    // per lane instead of 16 scattered 8-B ones (the row-per-lane epilogue is store-issue bound, guide T21).
    unsigned char* mine = smem + wave * (32 * 144);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *(uint2*)(mine + l31 * 144 + (d * 32 + 8 * q4 + 4 * hi) * 2) =
                pack4<P>(o[d][4 * q4 + 0] * inv, o[d][4 * q4 + 1] * inv, o[d][4 * q4 + 2] * inv, o[d][4 * q4 + 3] * inv);
    const int rsub = lane >> 3, seg = lane & 7;
    unsigned char* obase = (unsigned char*)a.out + (((size_t)n * T + qt * QB + wave * 32) * (H * 64) + h * 64) * 2 + seg * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + rsub;
        const uint4 v = *(const uint4*)(mine + row * 144 + seg * 16);
        if (qt * QB + wave * 32 + row < T) *(uint4*)(obase + (size_t)row * (H * 64) * 2) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Small grids (single-utterance synthesis: n_items * H * ceil(T/256) blocks of the kernel above would occupy a few
// CUs and walk their key tiles one after the other): one block = 32 queries of one (item, head), and its 8 waves
// split the KEY tiles between them (wave w takes tiles w, w+8, ...), each wave streaming its own K / V^T tile into
// its own 16 KB of LDS -- no block barrier in the main phase.  The 8 partial (m, l, O) triples are then merged
// through LDS with the usual log-sum-exp rescale.  Same operand layouts, same MFMA formulation as above.
constexpr int kAttnSmallLds = 8 * 2 * 64 * 128;
template <class P>
__global__ __launch_bounds__(512, 1) void attention_small_kernel(const AttnArgs a) {
    constexpr int NW = 8, TILE_BYTES = 64 * 128, WAVE_LDS = 2 * TILE_BYTES, OP = 33, QTILE = 32;
    using vec8 = typename P::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int T = a.T, Tp = a.Tp, H = a.H;
    const int qgroups = (T + 31) / 32;
    const int total = a.n_items * H * qgroups;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int qt = lin % qgroups;
    const int nh = lin / qgroups;
    const int n = nh / H, h = nh % H;
    const int mb = n % a.mask_mod;
    if (a.t_lim && qt * QTILE >= a.t_lim[mb]) return;        // ragged batch: every query of this tile is past the item's last needed frame
    const int kvend = a.kv_end[mb];
    const int nfull = a.n_full[mb];
    const float* kbias = a.kbias + (size_t)mb * Tp;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int query = qt * 32 + l31;

    const unsigned char* qbase = (const unsigned char*)a.q + ((size_t)nh * T) * 128;
    const unsigned char* kbase = (const unsigned char*)a.k + ((size_t)nh * T) * 128;
    const unsigned char* vbase = (const unsigned char*)a.vt + ((size_t)nh * 64) * Tp * 2;
    const unsigned char* zeros = (const unsigned char*)a.zeros;
    unsigned char* Ks = smem + wave * WAVE_LDS;
    unsigned char* Vs = Ks + TILE_BYTES;

    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (query < T) v = *(const uint4*)(qbase + (size_t)query * 128 + ks * 32 + hi * 16);
        qf[ks] = as_vec8<P>(v);
    }
    int row_off[2], swz[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int row = b * 32 + l31;
        row_off[b] = row * 128; swz[b] = (row >> 1) & 7;
    }
    f32x16_t o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles = (kvend + 63) >> 6;
    for (int kt = wave; kt < ntiles; kt += NW) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous tile's fragment reads have returned
#pragma unroll
        for (int piece = 0; piece < 8; ++piece) {
            const int row = piece * 8 + (lane >> 3);
            const int seg = (lane & 7) ^ ((row >> 1) & 7);
            const int key = kt * 64 + row;
            glds16b(key < T ? kbase + (size_t)key * 128 + seg * 16 : zeros, Ks + piece * 1024);
            glds16b(vbase + ((size_t)row * Tp + kt * 64 + seg * 8) * 2, Vs + piece * 1024);
        }
        ST_DMA_WAIT(0);

        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const unsigned char* kp = Ks + row_off[kb];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[kb] = P::mfma(as_vec8<P>(*(const uint4*)(kp + (((ks * 2 + hi) ^ swz[kb]) << 4))), qf[ks], s[kb]);
        }
        if ((kt + 1) * 64 > nfull) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bz = *(const float4*)(kbias + kt * 64 + kb * 32 + 8 * g4 + 4 * hi);
                    s[kb][4 * g4 + 0] += bz.x; s[kb][4 * g4 + 1] += bz.y;
                    s[kb][4 * g4 + 2] += bz.z; s[kb][4 * g4 + 3] += bz.w;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = xor32_max(mx);
        const float m_new = fmaxf(m_run, mx);
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            m_run = m_new;
        }
        vec8 pf[4];
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_run);
                psum += p;
                pf[kb * 2 + (r >> 3)][r & 7] = to16<P>(p);
            }
        l_run += psum;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const unsigned char* vp = Vs + row_off[d];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                o[d] = P::mfma(as_vec8<P>(*(const uint4*)(vp + (((g * 2 + hi) ^ swz[d]) << 4))), pf[g], o[d]);
        }
    }

    // ---- merge the 8 partial results: each wave parks (O^T [d][query] fp32, m, l) in its own LDS region
    const float l_tot = xor32_sum(l_run);
    float* ow = (float*)(smem + wave * WAVE_LDS);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ow[(d * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * OP + l31] = o[d][r];
    if (hi == 0) { ow[64 * OP + l31] = m_run; ow[64 * OP + 32 + l31] = l_tot; }
    __syncthreads();
    const int q = tid >> 4, d0 = (tid & 15) * 4;
    float mw[NW], M = -1e30f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { mw[w] = ((const float*)(smem + w * WAVE_LDS))[64 * OP + q]; M = fmaxf(M, mw[w]); }
    float L = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float* pw = (const float*)(smem + w * WAVE_LDS);
        const float sc = __builtin_amdgcn_exp2f(mw[w] - M);
        L += pw[64 * OP + 32 + q] * sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += pw[(d0 + i) * OP + q] * sc;
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    if (qt * 32 + q < T)
        *(uint2*)((unsigned char*)a.out + (((size_t)n * T + qt * 32 + q) * (H * 64) + h * 64 + d0) * 2) =
            pack4<P>(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
}

template <class P>
static hipError_t launch_attention_small(const AttnArgs& a, hipStream_t s) {
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_done_dev[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_small_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, kAttnSmallLds);
        if (e != hipSuccess) return e;
        attr_done_dev[dev_] = true;
    }
    const int total = a.n_items * a.H * ((a.T + 31) / 32);
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL((attention_small_kernel<P>), dim3(grid), dim3(512), kAttnSmallLds, s, a);
    return hipGetLastError();
}

hipError_t launch_attention(int dtype, const AttnArgs& a, hipStream_t s) {
    if (!a.zeros || !a.kbias) return hipErrorInvalidValue;
    const int qtiles = (a.T + 32 * ST_ATTN_WAVES - 1) / (32 * ST_ATTN_WAVES);
    const int total = a.n_items * a.H * qtiles;
    if (!a.lse && !a.q_lo && a.small_max_blocks > 0 && total <= a.small_max_blocks)
        return dtype == DT_BF16 ? launch_attention_small<OpBF16>(a, s) : launch_attention_small<OpF16>(a, s);
    const int grid = 8 * ((a.n_items * a.H + 7) / 8) * qtiles;      // 8 XCDs x (item, head) groups per XCD x query tiles
    if (a.lse) {
        if (a.vt_lo) {      // v as a hi + lo pair (72 KB of LDS, as the split-score kernel)
            if (dtype == DT_BF16) hipLaunchKernelGGL((attention_kernel<OpBF16, true, false, true>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
            else                  hipLaunchKernelGGL((attention_kernel<OpF16, true, false, true>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
            return hipGetLastError();
        }
        if (dtype == DT_BF16) hipLaunchKernelGGL((attention_kernel<OpBF16, true>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
        else                  hipLaunchKernelGGL((attention_kernel<OpF16, true>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
        return hipGetLastError();
    }
    if (a.q_lo || a.k_lo) {       // split-precision scores (72 KB of LDS: above the 64 KB default, opt in per device)
        if (!a.q_lo || !a.k_lo) return hipErrorInvalidValue;
        static bool attr_done_dev[64][2] = {};
        int dev_ = 0;
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
        const int di = dtype == DT_BF16 ? 0 : 1;
        const void* fn = di == 0 ? (const void*)attention_kernel<OpBF16, false, true> : (const void*)attention_kernel<OpF16, false, true>;
        if (!attr_done_dev[dev_][di]) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
            (void)e;
            attr_done_dev[dev_][di] = true;
        }
        if (di == 0) hipLaunchKernelGGL((attention_kernel<OpBF16, false, true>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
        else         hipLaunchKernelGGL((attention_kernel<OpF16, false, true>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
        return hipGetLastError();
    }
    if (dtype == DT_BF16) hipLaunchKernelGGL((attention_kernel<OpBF16, false>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
    else                  hipLaunchKernelGGL((attention_kernel<OpF16, false>), dim3(grid), dim3(64 * ST_ATTN_WAVES), 0, s, a);
    return hipGetLastError();
}

}  // namespace st
