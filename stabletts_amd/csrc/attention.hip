// Flash-style masked self-attention for gfx950 (replaces F.scaled_dot_product_attention with the
// materialised (B,1,T,T) additive mask of models/diffusion_transformer.py:77,107-108).
//
//   q, k : [item][H][T][64]   16-bit, RoPE applied; q pre-scaled by log2(e)/sqrt(64)
//   vT   : [item][H][64][Tp]  16-bit (keys contiguous; Tp = T rounded up to 64, zero tail)
//   out  : [item][T][H*64]    16-bit
//
// One block = 4 waves = 128 queries of one (item, head); each wave owns 32 queries.  K / V^T tiles
// of 64 keys are staged through LDS (register-staged, double-buffered, one barrier per tile).
// Both MFMAs are issued "transposed" so that a QUERY IS A LANE everywhere:
//   S^T[key][query] = K . Q^T      (A = K tile from LDS, B = Q fragment held in registers)
//   O^T[d][query]  += V^T . P^T    (A = V^T tile from LDS, B = P^T = exp2(S^T - m) in registers)
// The accumulator layout of S^T (lane = query, registers = keys) IS the B-operand layout of P^T up
// to a permutation of the k-slots, and a permutation of k-slots applied to both operands does not
// change a dot product -- so P never leaves registers and never crosses lanes; the V^T tile is
// written to LDS with key bits 2<->3 swapped to match.  Softmax max/sum are lane-local plus one
// shuffle with lane^32 (the other half of the same query's keys).
// Key masking comes from the per-utterance mask row, not from a T x T tensor: tiles below the
// valid prefix skip masking entirely, tiles past the last valid key are never visited.
// Padded QUERY rows produce finite garbage that the out-projection epilogue multiplies by 0,
// exactly as the reference's uniform-softmax rows are zeroed by "* x_mask" (:111).
#include "common.h"
#include "launch.h"

namespace st {

template <class P>
__global__ __launch_bounds__(256, 2) void attention_kernel(const AttnArgs a) {
    using vec8 = typename P::vec8;
    constexpr int ROWB = kLdsRowBytes;
    constexpr int TILE_BYTES = 64 * ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
    unsigned char* Ks = smem;                    // 2 buffers
    unsigned char* Vs = smem + 2 * TILE_BYTES;   // 2 buffers

    const int T = a.T, Tp = a.Tp, H = a.H;
    const int qtiles = (T + 127) >> 7;
    const int total = a.n_items * H * qtiles;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int qt = lin % qtiles;
    const int nh = lin / qtiles;
    const int n = nh / H, h = nh % H;
    const int mb = n % a.mask_mod;
    const int kvend = a.kv_end[mb];
    const int nfull = a.n_full[mb];
    const float* mrow = a.mask + (size_t)mb * T;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int query = qt * 128 + wave * 32 + l31;

    const unsigned char* qbase = (const unsigned char*)a.q + ((size_t)nh * T) * 128;
    const unsigned char* kbase = (const unsigned char*)a.k + ((size_t)nh * T) * 128;
    const unsigned char* vbase = (const unsigned char*)a.vt + ((size_t)nh * 64) * Tp * 2;

    // Q fragments (B operand): lane (query, hi) holds head dims ks*16 + hi*8 .. +8
    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (query < T) v = *(const uint4*)(qbase + (size_t)query * 128 + ks * 32 + hi * 16);
        qf[ks] = as_vec8<P>(v);
    }

    const int ntiles = (kvend + 63) >> 6;
    uint4 rk[2], rv[2];
    auto loadKV = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;
            const int row = idx >> 3, seg = idx & 7;
            const int key = kt * 64 + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (key < T) v = *(const uint4*)(kbase + (size_t)key * 128 + seg * 16);
            rk[i] = v;
            // V^T row = head dim `row`, 8 consecutive keys kt*64 + seg*8 .. +8 (always inside Tp)
            rv[i] = *(const uint4*)(vbase + ((size_t)row * Tp + kt * 64 + seg * 8) * 2);
        }
    };
    auto storeKV = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;
            const int row = idx >> 3, seg = idx & 7;
            *(uint4*)(Ks + buf * TILE_BYTES + row * ROWB + seg * 16) = rk[i];
            // key permutation inside each 16-key group: swap key bits 2 and 3
            unsigned char* vrow = Vs + buf * TILE_BYTES + row * ROWB + (seg >> 1) * 32 + (seg & 1) * 8;
            *(uint2*)(vrow) = make_uint2(rv[i].x, rv[i].y);        // keys +0..3  -> slot b3*4
            *(uint2*)(vrow + 16) = make_uint2(rv[i].z, rv[i].w);   // keys +4..7  -> slot 8 + b3*4
        }
    };

    f32x16_t o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    if (ntiles > 0) {
        loadKV(0);
        storeKV(0);
    }
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) loadKV(kt + 1);

        // ---- S^T = K . Q^T  (log2 units)
        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            const unsigned char* kp = Ks + buf * TILE_BYTES + (kb * 32 + l31) * ROWB + hi * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[kb] = P::mfma(as_vec8<P>(*(const uint4*)(kp + ks * 32)), qf[ks], s[kb]);
        }
        // ---- key mask (only tiles that are not entirely inside the valid prefix)
        if ((kt + 1) * 64 > nfull) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool valid = (key < T) && (mrow[key < T ? key : 0] != 0.0f);
                    s[kb][r] = valid ? s[kb][r] : -1e30f;
                }
        }
        // ---- online softmax, one query per lane (pair lane^32 shares the query)
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        vec8 pf[4];
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(s[kb][r] - m_new);
                psum += p;
                pf[kb * 2 + (r >> 3)][r & 7] = to16<P>(p);
            }
        l_run += psum;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const unsigned char* vp = Vs + buf * TILE_BYTES + (d * 32 + l31) * ROWB + hi * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                o[d] = P::mfma(as_vec8<P>(*(const uint4*)(vp + g * 32)), pf[g], o[d]);
        }

        if (kt + 1 < ntiles) storeKV(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (query < T) {
        unsigned char* dst = (unsigned char*)a.out + (((size_t)n * T + query) * (H * 64) + h * 64) * 2;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int dd = d * 32 + 8 * q4 + 4 * hi;
                *(uint2*)(dst + dd * 2) = pack4<P>(o[d][4 * q4 + 0] * inv, o[d][4 * q4 + 1] * inv,
                                                   o[d][4 * q4 + 2] * inv, o[d][4 * q4 + 3] * inv);
            }
    }
}

hipError_t launch_attention(int dtype, const AttnArgs& a, hipStream_t s) {
    const int qtiles = (a.T + 127) / 128;
    const int total = a.n_items * a.H * qtiles;
    const int grid = ((total + 7) / 8) * 8;
    if (dtype == DT_BF16) hipLaunchKernelGGL((attention_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, a);
    else                  hipLaunchKernelGGL((attention_kernel<OpF16>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace st
