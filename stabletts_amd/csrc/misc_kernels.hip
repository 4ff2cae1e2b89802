// Memory-bound glue kernels of the CFM decoder path (gfx950): FiLM + LayerNorm + adaLN modulate,
// layout changes at the drop-in boundary, time embedding, small fp32 linears, CFG combine and
// ODE state updates, weight packing.  All fp32 arithmetic; 16-byte vector accesses.
#include "common.h"
#include "launch.h"

namespace st {

// ------------------------------------------------------------------------------------------
// FiLM (estimator.py:31-33,16) + "* mask" + LayerNorm(C=256, eps 1e-5, no affine) + modulate
// (diffusion_transformer.py:111-112,119-121).  One wave per frame: lane holds 4 channels.
template <class P>
__global__ __launch_bounds__(256) void film_ln_kernel(const FilmLnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < a.rows; row += gridDim.x * 4) {
        const int n = row / a.T, t = row - n * a.T;
        const float m = a.mask ? a.mask[(size_t)(n % a.mask_mod) * a.T + t] : 1.0f;
        float4 x = *(const float4*)(a.X + (size_t)row * 256 + lane * 4);
        if (a.film) {
            const float* f = a.film + (size_t)(n % a.film_mod) * a.film_stride + lane * 4;
            const float4 ga = *(const float4*)f;
            const float4 be = *(const float4*)(f + 256);
            x.x = (ga.x * x.x + be.x) * m; x.y = (ga.y * x.y + be.y) * m;
            x.z = (ga.z * x.z + be.z) * m; x.w = (ga.w * x.w + be.w) * m;
            *(float4*)(a.X + (size_t)row * 256 + lane * 4) = x;
        }
        const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / 256.0f);
        const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
        const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float* ad = a.ada + (size_t)n * a.ada_stride + lane * 4;
        const float4 sh = *(const float4*)(ad + a.shift_off);
        const float4 sc = *(const float4*)(ad + a.scale_off);
        float h0 = d0 * rstd * (1.0f + sc.x) + sh.x;
        float h1 = d1 * rstd * (1.0f + sc.y) + sh.y;
        float h2 = d2 * rstd * (1.0f + sc.z) + sh.z;
        float h3 = d3 * rstd * (1.0f + sc.w) + sh.w;
        if (a.mask_out) { h0 *= m; h1 *= m; h2 *= m; h3 *= m; }
        *(uint2*)((unsigned char*)a.h16 + ((size_t)row * 256 + lane * 4) * 2) = pack4<P>(h0, h1, h2, h3);
    }
}

hipError_t launch_film_ln(int dtype, const FilmLnArgs& a, hipStream_t s) {
    int grid = (a.rows + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
    if (grid < 1) grid = 1;
    if (dtype == DT_BF16) hipLaunchKernelGGL((film_ln_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, a);
    else                  hipLaunchKernelGGL((film_ln_kernel<OpF16>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// SinusoidalPosEmb (estimator.py:41-49): emb[i] = sin(1000 t f_i), emb[half+i] = cos(...),
// f_i = exp(-i ln(1e4)/(half-1)).
__global__ void time_embed_kernel(const float* t, int n_t, int dim, float* emb) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_t * half) return;
    const int n = idx / half, i = idx - n * half;
    const float e = logf(10000.0f) / (float)(half - 1);
    const float f = expf((float)i * -e);
    const float arg = 1000.0f * t[n] * f;
    emb[(size_t)n * dim + i] = sinf(arg);
    emb[(size_t)n * dim + half + i] = cosf(arg);
}

hipError_t launch_time_embed(const float* t, int n_t, int dim, float* emb, hipStream_t s) {
    const int total = n_t * (dim / 2);
    hipLaunchKernelGGL(time_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, n_t, dim, emb);
    return hipGetLastError();
}

// small dense layer, one wave per output element; k % 4 == 0
__global__ __launch_bounds__(256) void linear_kernel(const float* in, int n, int k, const float* W, const float* bias,
                                                     int o, float* out, int silu_in, int silu_out) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)n * o) return;
    const int ni = (int)(wid / o), oi = (int)(wid - (long long)ni * o);
    const float* x = in + (size_t)ni * k;
    const float* w = W + (size_t)oi * k;
    float acc = 0.f;
    for (int i = lane * 4; i < k; i += 256) {
        float4 xv = *(const float4*)(x + i);
        const float4 wv = *(const float4*)(w + i);
        if (silu_in) { xv.x = silu_f(xv.x); xv.y = silu_f(xv.y); xv.z = silu_f(xv.z); xv.w = silu_f(xv.w); }
        acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float v = acc + (bias ? bias[oi] : 0.f);
        if (silu_out) v = silu_f(v);
        out[(size_t)ni * o + oi] = v;
    }
}

__global__ __launch_bounds__(256) void silu_rows_kernel(const float* in, int64_t n, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = silu_f(in[i]);
}
// out = SiLU(in): the activation in front of the six adaLN modulation linears is the same vector for all of them; computed
// once instead of once per output element inside linear_kernel (its expf chain was most of those launches' 14 us)
hipError_t launch_silu_rows(const float* in, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(silu_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, out);
    return hipGetLastError();
}

// the same layer shape for up to 8 (weight, bias, output[, input]) sets in ONE launch (grid.y = set): the six FiLM linears of a training
// forward share their input tau, the six adaLN modulation linears c -- were 12 launches of ~10 us.
// Block = 16 outputs x 32 items (grid.z walks the item groups): the 32 input rows and the 16 weight rows go through LDS in k-chunks of
// 256, thread (item = tid & 31, output pair = tid >> 5) keeps two accumulators; every weight element is read from memory once per item
// group instead of once per item (one wave per output element took 136 us for 6 x [64 x 256] -> 1536).
__global__ __launch_bounds__(256) void linear_multi_kernel(LinearJobs J, int n, int k, int o, int silu_in, int silu_out) {
    constexpr int NI = 32, OT = 16, KC = 256, XP = KC + 4;
    __shared__ __attribute__((aligned(16))) float xs[NI][XP];
    __shared__ __attribute__((aligned(16))) float ws[OT][KC];
    const int tid = threadIdx.x;
    const int set = blockIdx.y, o0 = blockIdx.x * OT, n0 = blockIdx.z * NI;
    const float* X = J.in[set];
    const float* W = J.W[set];
    const int ni = tid & 31, op = tid >> 5;
    float acc0 = 0.f, acc1 = 0.f;
    for (int k0 = 0; k0 < k; k0 += KC) {
        const int kc = min(KC, k - k0);          // multiple of 4
#pragma unroll
        for (int i = 0; i < NI * KC / 4 / 256; ++i) {
            const int idx = tid + 256 * i, r = idx / (KC / 4), c = (idx % (KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + r < n && c < kc) {
                v = *(const float4*)(X + (size_t)(n0 + r) * k + k0 + c);
                if (silu_in) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            }
            *(float4*)&xs[r][c] = v;
        }
#pragma unroll
        for (int i = 0; i < OT * KC / 4 / 256; ++i) {
            const int idx = tid + 256 * i, r = idx / (KC / 4), c = (idx % (KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o0 + r < o && c < kc) v = *(const float4*)(W + (size_t)(o0 + r) * k + k0 + c);
            *(float4*)&ws[r][c] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < KC; c += 4) {
            const float4 xv = *(const float4*)&xs[ni][c];
            const float4 w0 = *(const float4*)&ws[2 * op][c];
            const float4 w1 = *(const float4*)&ws[2 * op + 1][c];
            acc0 += xv.x * w0.x + xv.y * w0.y + xv.z * w0.z + xv.w * w0.w;
            acc1 += xv.x * w1.x + xv.y * w1.y + xv.z * w1.z + xv.w * w1.w;
        }
        __syncthreads();
    }
    if (n0 + ni < n) {
        const float* bias = J.bias[set];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oi = o0 + 2 * op + j;
            if (oi < o) {
                float v = (j ? acc1 : acc0) + (bias ? bias[oi] : 0.f);
                if (silu_out) v = silu_f(v);
                J.out[set][(size_t)(n0 + ni) * o + oi] = v;
            }
        }
    }
}
hipError_t launch_linear_multi(const LinearJobs& J, int n, int k, int o, int silu_in, int silu_out, hipStream_t s) {
    if (J.n < 1 || J.n > 8 || (k & 3) || n < 1 || o < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(linear_multi_kernel, dim3((unsigned)((o + 15) / 16), J.n, (unsigned)((n + 31) / 32)), dim3(256), 0, s, J, n, k, o, silu_in, silu_out);
    return hipGetLastError();
}

hipError_t launch_linear(const float* in, int n, int k, const float* W, const float* bias, int o,
                         float* out, int silu_in, int silu_out, hipStream_t s) {
    const long long waves = (long long)n * o;
    const int grid = (int)((waves + 3) / 4);
    hipLaunchKernelGGL(linear_kernel, dim3(grid), dim3(256), 0, s, in, n, k, W, bias, o, out, silu_in, silu_out);
    return hipGetLastError();
}

// per mask row: n_full = length of the leading run of non-zeros, kv_end = last non-zero + 1,
// kbias[t] = 0 for valid keys, -1e30 for masked keys and for t in [T, Tp)
__global__ __launch_bounds__(256) void mask_prep_kernel(const float* mask, int T, int Tp, int* n_full, int* kv_end,
                                                        float* kbias, int* t_lim) {
    __shared__ int s_first_zero, s_last_nz;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { s_first_zero = T; s_last_nz = -1; }
    __syncthreads();
    int fz = T, ln = -1;
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) {
        const bool nz = (t < T) && (mask[(size_t)b * T + t] != 0.0f);
        kbias[(size_t)b * Tp + t] = nz ? 0.0f : -1e30f;
        if (t < T) {
            if (!nz && t < fz) fz = t;
            if (nz && t > ln) ln = t;
        }
    }
    atomicMin(&s_first_zero, fz);
    atomicMax(&s_last_nz, ln);
    __syncthreads();
    if (threadIdx.x == 0) {
        n_full[b] = s_first_zero; kv_end[b] = s_last_nz + 1;
        if (t_lim) {
            const int lim = min(T, s_last_nz + 1 + kFrameHalo);
            t_lim[b] = lim;
            atomicMax(&t_lim[gridDim.x], lim);      // entry B: the longest row (zeroed by the launcher)
        }
    }
}

hipError_t launch_mask_prep(const float* mask, int B, int T, int Tp, int* n_full, int* kv_end, float* kbias, int* t_lim, hipStream_t s) {
    if (t_lim) { hipError_t e = hipMemsetAsync(t_lim + B, 0, sizeof(int), s); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(mask_prep_kernel, dim3(B), dim3(256), 0, s, mask, T, Tp, n_full, kv_end, kbias, t_lim);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// boundary layout changes: (B, C, T) <-> time-major (B, T, Cp), 32x32 tiles through LDS
template <class P>
__global__ __launch_bounds__(256) void to_time_major_kernel(const float* in, int C, int T, int Cp,
                                                            float* out32, typename P::elem* out16, typename P::elem* out16lo) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, t = t0 + tx;
        float v = 0.f;
        if (c < C && t < T) v = in[((size_t)b * C + c) * T + t];
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + i * 8, c = c0 + tx;
        if (t < T && c < Cp) {
            const float v = tile[tx][ty + i * 8];
            const size_t o = ((size_t)b * T + t) * Cp + c;
            if (out32) out32[o] = v;
            if (out16) out16[o] = to16<P>(v);
            if (out16lo) out16lo[o] = to16<P>(v - (float)to16<P>(v));
        }
    }
}

hipError_t launch_to_time_major(int dtype, const float* in, int B, int C, int T, int Cp,
                                float* out32, void* out16, void* out16lo, hipStream_t s) {
    dim3 grid((T + 31) / 32, (Cp + 31) / 32, B);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((to_time_major_kernel<OpBF16>), grid, dim3(256), 0, s, in, C, T, Cp, out32, (__bf16*)out16, (__bf16*)out16lo);
    else
        hipLaunchKernelGGL((to_time_major_kernel<OpF16>), grid, dim3(256), 0, s, in, C, T, Cp, out32, (_Float16*)out16, (_Float16*)out16lo);
    return hipGetLastError();
}

// nonfinite (optional, host-mapped): set to 1 when a value that leaves through this boundary is NaN / Inf (an f16 operand that
// overflowed, a bad input) -- read by st_output_status and by the next call's arena check, never waited for here
__global__ __launch_bounds__(256) void from_time_major_kernel(const float* in, int C, int T, int Cp, float* out, int* nonfinite) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + i * 8, c = c0 + tx;
        float v = 0.f;
        if (t < T && c < Cp) v = in[((size_t)b * T + t) * Cp + c];
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, t = t0 + tx;
        if (c < C && t < T) {
            const float v = tile[tx][ty + i * 8];
            out[((size_t)b * C + c) * T + t] = v;
            if (nonfinite && !(fabsf(v) <= 3.4028234664e38f)) *nonfinite = 1;
        }
    }
}

hipError_t launch_from_time_major(const float* in, int B, int C, int T, int Cp, float* out, hipStream_t s, int* nonfinite) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(from_time_major_kernel, grid, dim3(256), 0, s, in, C, T, Cp, out, nonfinite);
    return hipGetLastError();
}

template <class P>
__global__ void fill_rows16_kernel(const float* vec, int C, int Cp, int T, typename P::elem* out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)T * Cp) return;
    const int c = (int)(idx % Cp);
    out[idx] = to16<P>(c < C ? vec[c] : 0.f);
}

hipError_t launch_fill_rows16(int dtype, const float* vec, int C, int Cp, int T, void* out16, hipStream_t s) {
    const size_t total = (size_t)T * Cp;
    const int grid = (int)((total + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((fill_rows16_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, vec, C, Cp, T, (__bf16*)out16);
    else                  hipLaunchKernelGGL((fill_rows16_kernel<OpF16>), dim3(grid), dim3(256), 0, s, vec, C, Cp, T, (_Float16*)out16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Small per-solve setup helpers (no D2D memcpy calls, no host synchronisation on the solve path)
struct SetValuesArgs { float v[64]; };
__global__ void set_values_kernel(float* dst, int n, SetValuesArgs a) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = a.v[i];
}
// host values -> device array through kernel arguments (64 per launch): the evaluation times of a solve
// holds its stream for about `us` microseconds (one wave sleeping; 100-MHz constant-rate counter): the phase offset between the
// launch sequences of a multi-part solve (engine.cpp)
__global__ void delay_kernel(unsigned long long ticks, int max_iters) {
    // s_memrealtime counts the constant-rate wall clock (hipDeviceAttributeWallClockRate, 100 MHz on MI355X); the iteration cap bounds the
    // wait should the rate be different on another ASIC / driver setting (s_sleep 16 = 1024 cycles: <= ~1 us per iteration at any clock)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < max_iters && __builtin_amdgcn_s_memrealtime() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}
hipError_t launch_delay(int us, hipStream_t s) {
    static int khz[64] = {};      // wall-clock rate per device, queried once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!khz[dev]) {
        int r = 0;
        if (hipDeviceGetAttribute(&r, hipDeviceAttributeWallClockRate, dev) != hipSuccess || r <= 0) { (void)hipGetLastError(); r = 100000; }
        khz[dev] = r;
    }
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, (unsigned long long)us * (unsigned long long)khz[dev] / 1000ull, 4 * us + 64);
    return hipGetLastError();
}

hipError_t launch_set_values(float* dst, const float* host_vals, int n, hipStream_t s) {
    for (int o = 0; o < n; o += 64) {
        SetValuesArgs a;
        const int m = n - o < 64 ? n - o : 64;
        for (int i = 0; i < 64; ++i) a.v[i] = i < m ? host_vals[o + i] : 0.f;
        hipLaunchKernelGGL(set_values_kernel, dim3(1), dim3(64), 0, s, dst + o, m, a);
    }
    return hipGetLastError();
}

// speaker rows of a solve: rows [0, B) = c, rows [B, 2B) = the CFG null speaker repeated (flow_matching.py:60)
__global__ void cvec_prep_kernel(const float* c, const float* fake, int B, int G, float* dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = B * G;
    if (i < n) dst[i] = c[i];
    else if (fake && i < 2 * n) dst[i] = fake[(i - n) % G];
}
hipError_t launch_cvec_prep(const float* c, const float* fake_or_null, int B, int G, float* dst, hipStream_t s) {
    const int total = (fake_or_null ? 2 : 1) * B * G;
    hipLaunchKernelGGL(cvec_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, s, c, fake_or_null, B, G, dst);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// TextEncoder front end (models/text_encoder.py:35-37): x = emb[token] * sqrt(C), time-major fp32, and the
// sequence mask (utils/mask.py: t < length) in both the (B,1,T) boundary layout and as the engine's mask row.
// The block's first statement x = x * x_mask (diffusion_transformer.py:106) is applied here.
// One wave per (item, position) row; out-of-range token ids are clamped to [0, n_vocab).
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long long* tokens, const long long* lengths,
                                                           const float* emb, int n_vocab, int C, float scale, int B,
                                                           int T, float* X, float* mask_out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * T) return;
    const int b = row / T, t = row % T;
    const float m = (long long)t < lengths[b] ? 1.0f : 0.0f;
    long long tok = tokens[row];
    tok = tok < 0 ? 0 : (tok >= n_vocab ? n_vocab - 1 : tok);
    const float s = scale * m;
    for (int ch = lane * 4; ch < C; ch += 256) {
        const float4 v = *(const float4*)(emb + (size_t)tok * C + ch);
        *(float4*)(X + (size_t)row * C + ch) = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    }
    if (lane == 0) mask_out[row] = m;
}
hipError_t launch_embed_tokens(const long long* tokens, const long long* lengths, const float* emb, int n_vocab,
                               int C, float scale, int B, int T, float* X, float* mask_out, hipStream_t s) {
    const int rows = B * T;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, tokens, lengths, emb, n_vocab, C,
                       scale, B, T, X, mask_out);
    return hipGetLastError();
}

// rounding residual of a 4-vector against its 16-bit image: the "lo" half of a split-precision MFMA operand
template <class P>
__device__ __forceinline__ uint2 pack4_lo(float a, float b, float c, float d) {
    return pack4<P>(a - (float)to16<P>(a), b - (float)to16<P>(b), c - (float)to16<P>(c), d - (float)to16<P>(d));
}

// ------------------------------------------------------------------------------------------
// CFG combine (flow_matching.py:66) + optional fused Euler update (torchdiffeq euler step)
template <class P>
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* v, int64_t half, int use_cfg, float s,
                                                          float* kout, float* xio, typename P::elem* x16,
                                                          typename P::elem* x16lo, float dt) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= half) return;
    float4 vc = *(const float4*)(v + i);
    if (use_cfg) {
        const float4 vu = *(const float4*)(v + half + i);
        vc.x = vu.x + s * (vc.x - vu.x); vc.y = vu.y + s * (vc.y - vu.y);
        vc.z = vu.z + s * (vc.z - vu.z); vc.w = vu.w + s * (vc.w - vu.w);
    }
    if (kout) *(float4*)(kout + i) = vc;
    if (xio) {
        float4 x = *(const float4*)(xio + i);
        x.x += dt * vc.x; x.y += dt * vc.y; x.z += dt * vc.z; x.w += dt * vc.w;
        *(float4*)(xio + i) = x;
        if (x16) *(uint2*)(x16 + i) = pack4<P>(x.x, x.y, x.z, x.w);
        if (x16lo) *(uint2*)(x16lo + i) = pack4_lo<P>(x.x, x.y, x.z, x.w);
    }
}

hipError_t launch_cfg_combine(int dtype, const float* v, int B, int64_t per_item, int use_cfg, float s,
                              float* kout, float* xio, void* x16, void* x16lo, float dt, hipStream_t stream) {
    const int64_t half = (int64_t)B * per_item;   // multiple of 4
    const int grid = (int)((half / 4 + 255) / 256);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((cfg_combine_kernel<OpBF16>), dim3(grid), dim3(256), 0, stream, v, half, use_cfg, s, kout, xio, (__bf16*)x16, (__bf16*)x16lo, dt);
    else
        hipLaunchKernelGGL((cfg_combine_kernel<OpF16>), dim3(grid), dim3(256), 0, stream, v, half, use_cfg, s, kout, xio, (_Float16*)x16, (_Float16*)x16lo, dt);
    return hipGetLastError();
}

constexpr int kMaxComb = 7;   // dopri5 has 7 stage derivatives
struct LinCombArgs { const float* x; const float* k[kMaxComb]; float coef[kMaxComb]; int nk; int64_t n; float* y32; void* y16; void* y16lo; };

template <class P>
__global__ __launch_bounds__(256) void lincomb_kernel(const LinCombArgs a) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= a.n) return;
    float4 y = *(const float4*)(a.x + i);
#pragma unroll
    for (int j = 0; j < kMaxComb; ++j) {
        if (j < a.nk) {
            const float4 kv = *(const float4*)(a.k[j] + i);
            const float c = a.coef[j];
            y.x += c * kv.x; y.y += c * kv.y; y.z += c * kv.z; y.w += c * kv.w;
        }
    }
    if (a.y32) *(float4*)(a.y32 + i) = y;
    if (a.y16) *(uint2*)((typename P::elem*)a.y16 + i) = pack4<P>(y.x, y.y, y.z, y.w);
    if (a.y16lo) *(uint2*)((typename P::elem*)a.y16lo + i) = pack4_lo<P>(y.x, y.y, y.z, y.w);
}

hipError_t launch_lincomb(int dtype, const float* x, const float* const* k, const float* coef, int nk,
                          int64_t n, float* y32, void* y16, void* y16lo, hipStream_t s) {
    LinCombArgs a;
    a.x = x; a.nk = nk; a.n = n; a.y32 = y32; a.y16 = y16; a.y16lo = y16lo;
    for (int j = 0; j < kMaxComb; ++j) { a.k[j] = j < nk ? k[j] : nullptr; a.coef[j] = j < nk ? coef[j] : 0.f; }
    const int grid = (int)((n / 4 + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((lincomb_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, a);
    else                  hipLaunchKernelGGL((lincomb_kernel<OpF16>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// weight packing: (cout, cin_total, K) fp32 -> [row_off + co][K][cin_p] 16-bit, columns [col_off, col_off + slice_w)
template <class P>
__global__ void pack_weight_kernel(const float* src, int cout, int cin_total, int K, int ci_off, int ci_cnt,
                                   typename P::elem* dst, int row_off, int cin_p, int col_off, int slice_w, int lo) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)cout * K * slice_w;
    if (idx >= total) return;
    const int ci = (int)(idx % slice_w);
    const int j = (int)((idx / slice_w) % K);
    const int co = (int)(idx / ((size_t)slice_w * K));
    float v = 0.f;
    if (ci < ci_cnt) v = src[((size_t)co * cin_total + ci_off + ci) * K + j];
    if (lo) v -= (float)to16<P>(v);
    dst[((size_t)(row_off + co) * K + j) * cin_p + col_off + ci] = to16<P>(v);
}

hipError_t launch_pack_weight(int dtype, const float* src, int cout, int cin_total, int K, int ci_off,
                              int ci_cnt, void* dst, int row_off, int cin_p, int col_off, int slice_w, int lo,
                              hipStream_t s) {
    const size_t total = (size_t)cout * K * slice_w;
    const int grid = (int)((total + 255) / 256);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((pack_weight_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, src, cout, cin_total, K, ci_off, ci_cnt, (__bf16*)dst, row_off, cin_p, col_off, slice_w, lo);
    else
        hipLaunchKernelGGL((pack_weight_kernel<OpF16>), dim3(grid), dim3(256), 0, s, src, cout, cin_total, K, ci_off, ci_cnt, (_Float16*)dst, row_off, cin_p, col_off, slice_w, lo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// CFMDecoder.compute_loss's own arithmetic (models/flow_matching.py:86-100) around the estimator call: the interpolant and the
// target (prep), the masked-sum MSE (loss) and its gradient (loss_bwd).  Boundary layout (B, M, T) fp32, elementwise.
__global__ __launch_bounds__(256) void cfm_loss_prep_kernel(const float* x1, const float* z, const float* t_rand, float one_minus_sigma,
                                                            int B, int64_t per_item, float* t_out, float* y, float* u) {
    const int b = blockIdx.y;
    // t = 1 - cos(t_rand * 0.5 * pi) (:88), fp32 like the reference's tensor arithmetic
    const float t = 1.0f - cosf(t_rand[b] * 0.5f * 3.14159265358979323846f);
    if (blockIdx.x == 0 && threadIdx.x == 0) t_out[b] = t;
    const float cz = 1.0f - one_minus_sigma * t;      // y = (1 - (1 - sigma) t) z + t x1 (:93), u = x1 - (1 - sigma) z (:96)
    const int64_t base = (int64_t)b * per_item;
    // float4 path only where all four ADDRESSES are 16-byte aligned (the tensors may be views at any element offset)
    const bool al16 = ((((uintptr_t)(x1 + base)) | ((uintptr_t)(z + base)) | ((uintptr_t)(y + base)) | ((uintptr_t)(u + base))) & 15) == 0;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < per_item; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= per_item && al16) {
            const float4 a = *(const float4*)(x1 + base + i), n = *(const float4*)(z + base + i);
            *(float4*)(y + base + i) = make_float4(cz * n.x + t * a.x, cz * n.y + t * a.y, cz * n.z + t * a.z, cz * n.w + t * a.w);
            *(float4*)(u + base + i) = make_float4(a.x - one_minus_sigma * n.x, a.y - one_minus_sigma * n.y, a.z - one_minus_sigma * n.z, a.w - one_minus_sigma * n.w);
        } else {
            for (int64_t j = i; j < per_item && j < i + 4; ++j) {
                y[base + j] = cz * z[base + j] + t * x1[base + j];
                u[base + j] = x1[base + j] - one_minus_sigma * z[base + j];
            }
        }
    }
}
hipError_t launch_cfm_loss_prep(const float* x1, const float* z, const float* t_rand, float sigma_min, int B, int M, int T,
                                float* t_out, float* y, float* u, hipStream_t s) {
    const int64_t per_item = (int64_t)M * T;
    int gx = (int)((per_item / 4 + 255) / 256); if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(cfm_loss_prep_kernel, dim3(gx, B), dim3(256), 0, s, x1, z, t_rand, 1.0f - sigma_min, B, per_item, t_out, y, u);
    return hipGetLastError();
}

// sum (pred - u)^2 over every element and sum(mask): fixed partition, fixed combination order (deterministic).
// scratch: [0, kCfmLossBlocks) partial sums, [kCfmLossBlocks, 2 kCfmLossBlocks) partial mask sums, [2 k] = the squared sum,
// [2 k + 1] = the denominator sum(mask) * M (kept for the backward)
__global__ __launch_bounds__(256) void cfm_loss_partial_kernel(const float* pred, const float* u, const float* mask, int64_t n, int64_t nm, float* scratch) {
    __shared__ float red[2][4];
    float acc = 0.f, am = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - u[i];
        acc += d * d;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += (int64_t)gridDim.x * blockDim.x) am += mask[i];
    acc = wave_sum(acc); am = wave_sum(am);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = acc; red[1][threadIdx.x >> 6] = am; }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        scratch[kCfmLossBlocks + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ __launch_bounds__(256) void cfm_loss_final_kernel(float* scratch, int nblocks, int M, float* loss) {
    __shared__ float red[2][4];
    float a = 0.f, m = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += scratch[i]; m += scratch[kCfmLossBlocks + i]; }
    a = wave_sum(a); m = wave_sum(m);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ss = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float den = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (float)M;
        scratch[2 * kCfmLossBlocks] = ss; scratch[2 * kCfmLossBlocks + 1] = den;
        *loss = ss / den;          // mse_loss(pred, u, "sum") / (sum(mask) * n_feats) (:100); u is NOT masked (reference quirk)
    }
}
hipError_t launch_cfm_loss(const float* pred, const float* u, const float* mask, int B, int M, int T, float* scratch, float* loss, hipStream_t s) {
    const int64_t n = (int64_t)B * M * T, nm = (int64_t)B * T;
    int grid = (int)((n / 8 + 255) / 256); if (grid > kCfmLossBlocks) grid = kCfmLossBlocks; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cfm_loss_partial_kernel, dim3(grid), dim3(256), 0, s, pred, u, mask, n, nm, scratch);
    hipLaunchKernelGGL(cfm_loss_final_kernel, dim3(1), dim3(256), 0, s, scratch, grid, M, loss);
    return hipGetLastError();
}
// d loss / d pred = grad_loss * 2 (pred - u) / den
__global__ __launch_bounds__(256) void cfm_loss_bwd_kernel(const float* pred, const float* u, const float* scratch, const float* grad_loss, int64_t n, float* gpred) {
    const float f = 2.0f * grad_loss[0] / scratch[2 * kCfmLossBlocks + 1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) gpred[i] = f * (pred[i] - u[i]);
}
hipError_t launch_cfm_loss_bwd(const float* pred, const float* u, const float* scratch, const float* grad_loss, int B, int M, int T, float* gpred, hipStream_t s) {
    const int64_t n = (int64_t)B * M * T;
    int grid = (int)((n / 4 + 255) / 256); if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cfm_loss_bwd_kernel, dim3(grid), dim3(256), 0, s, pred, u, scratch, grad_loss, n, gpred);
    return hipGetLastError();
}

// fused-FFN weight stream (common.h: ffn_stream_index)
template <class P>
__global__ void pack_ffn_stream_kernel(const float* src, int stage, int F, typename P::elem* dst) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)F * 256 * 3) return;
    size_t so, dof;
    ffn_stream_index(idx, stage, F, &so, &dof);
    dst[dof] = to16<P>(src[so]);
}

hipError_t launch_pack_ffn_stream(int dtype, const float* src, int stage, int F, void* dst, hipStream_t s) {
    const size_t total = (size_t)F * 256 * 3;
    const int grid = (int)((total + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((pack_ffn_stream_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, src, stage, F, (__bf16*)dst);
    else                  hipLaunchKernelGGL((pack_ffn_stream_kernel<OpF16>), dim3(grid), dim3(256), 0, s, src, stage, F, (_Float16*)dst);
    return hipGetLastError();
}

// weight stream of the Winograd fused FFN (common.h: ffn_wino_index; f16 only)
__global__ void pack_ffn_wino_kernel(const float* src, int stage, int F, _Float16* dst) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)F * 256 * 3) return;
    size_t so, dof; int pl;
    ffn_wino_index(idx, stage, F, &so, &dof, &pl);
    dst[dof] = (_Float16)ffn_wino_plane(src + so, pl);
}
hipError_t launch_pack_ffn_wino(const float* src, int stage, int F, void* dst, hipStream_t s) {
    const size_t total = (size_t)F * 256 * 3;
    hipLaunchKernelGGL(pack_ffn_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, stage, F, (_Float16*)dst);
    return hipGetLastError();
}

// fragment-ordered q/k/v weight (common.h: qkv_frag_index)
template <class P>
__global__ void pack_qkv_frag_kernel(const float* src, int plane, typename P::elem* dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 256 * 256) return;
    dst[qkv_frag_index(plane, idx >> 8, idx & 255)] = to16<P>(src[idx]);
}
hipError_t launch_pack_qkv_frag(int dtype, const float* src, int plane, void* dst, hipStream_t s) {
    if (dtype == DT_BF16) hipLaunchKernelGGL((pack_qkv_frag_kernel<OpBF16>), dim3(256), dim3(256), 0, s, src, plane, (__bf16*)dst);
    else                  hipLaunchKernelGGL((pack_qkv_frag_kernel<OpF16>), dim3(256), dim3(256), 0, s, src, plane, (_Float16*)dst);
    return hipGetLastError();
}

template <class P>
__global__ void cvt16_kernel(const typename P::elem* src, float* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

hipError_t launch_cvt16_to_f32(int dtype, const void* src, float* dst, int64_t n, hipStream_t s) {
    const int grid = (int)((n + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((cvt16_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, (const __bf16*)src, dst, n);
    else                  hipLaunchKernelGGL((cvt16_kernel<OpF16>), dim3(grid), dim3(256), 0, s, (const _Float16*)src, dst, n);
    return hipGetLastError();
}

}  // namespace st
