// Training-only memory-bound kernels (gfx950): forward pieces that keep activations, backward of FiLM /
// LayerNorm+adaLN / gated residual / SiLU+dropout, per-item channel reductions, the operand transposes that
// turn the weight gradient into a call of the forward implicit-GEMM kernel, dgrad weight packing and the small
// fp32 linears.  All fp32 arithmetic; 16-bit tensors are MFMA operands only.
#include "common.h"
#include "train_launch.h"
#include <cstring>

namespace st {

// ------------------------------------------------------------------------------------------ dropout configuration
DropCfg make_drop(float p, unsigned long long seed, int salt) {
    DropCfg d;
    d.seed = seed * 0x100000001B3ull + (unsigned long long)(salt + 1) * 0xD6E8FEB86659FD93ull;
    d.rowh = nullptr; d.colh = nullptr;
    if (!(p > 0.f)) { d.thresh16 = 0; d.scale = 1.0f; return d; }
    const double t = (double)p * 65536.0 + 0.5;
    d.thresh16 = t >= 65535.0 ? 65535u : (unsigned)t;
    if (d.thresh16 == 0) d.thresh16 = 1;
    d.scale = 1.0f / (1.0f - p);
    return d;
}

__global__ __launch_bounds__(256) void drop_tables_kernel(unsigned long long seed, int n_rows, int n_colpairs, unsigned* rowh, unsigned* colh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) rowh[i] = drop_rowh(seed, (unsigned)i);
    if (i < n_colpairs) colh[i] = drop_colh(seed, (unsigned)i);
}
// ... for every attention site of one forward in ONE launch (grid.y = site); the backward re-reads the tables
__global__ __launch_bounds__(256) void drop_tables_multi_kernel(DropSeeds sd, int n_rows, int n_colpairs, unsigned* rowh, unsigned* colh,
                                                                size_t row_stride, size_t col_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long seed = sd.seed[blockIdx.y];
    if (i < n_rows) rowh[blockIdx.y * row_stride + i] = drop_rowh(seed, (unsigned)i);
    if (i < n_colpairs) colh[blockIdx.y * col_stride + i] = drop_colh(seed, (unsigned)i);
}
hipError_t launch_drop_tables_multi(const DropSeeds& sd, int n_sites, int n_rows, int n_colpairs, unsigned* rowh, unsigned* colh,
                                    size_t row_stride, size_t col_stride, hipStream_t s) {
    if (n_sites < 1 || n_sites > 16) return hipErrorInvalidValue;
    const int n = n_rows > n_colpairs ? n_rows : n_colpairs;
    hipLaunchKernelGGL(drop_tables_multi_kernel, dim3((n + 255) / 256, n_sites), dim3(256), 0, s, sd, n_rows, n_colpairs, rowh, colh, row_stride, col_stride);
    return hipGetLastError();
}
hipError_t launch_drop_tables(const DropCfg& d, int n_rows, int n_colpairs, unsigned* rowh, unsigned* colh, hipStream_t s) {
    const int n = n_rows > n_colpairs ? n_rows : n_colpairs;
    hipLaunchKernelGGL(drop_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d.seed, n_rows, n_colpairs, rowh, colh);
    return hipGetLastError();
}


template <class P>
__device__ __forceinline__ float4 load4_16(const void* p) {
    typedef __attribute__((ext_vector_type(4))) typename P::elem v4;
    const v4 v = *(const v4*)p;
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// ------------------------------------------------------------------------------------------ forward: FiLM / residual / LayerNorm
template <class P>
__global__ __launch_bounds__(256) void train_ln_kernel(const TrainLnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < a.rows; row += gridDim.x * 4) {
        const int n = row / a.T, t = row - n * a.T;
        const float m = a.mask ? a.mask[(size_t)(n % a.mask_mod) * a.T + t] : 1.0f;
        const size_t o = (size_t)row * 256 + lane * 4;
        float4 x = *(const float4*)(a.xin + o);
        if (a.film) {
            const float* f = a.film + (size_t)(n % a.film_mod) * a.film_stride + lane * 4;
            const float4 ga = *(const float4*)f, be = *(const float4*)(f + 256);
            x.x = (ga.x * x.x + be.x) * m; x.y = (ga.y * x.y + be.y) * m;
            x.z = (ga.z * x.z + be.z) * m; x.w = (ga.w * x.w + be.w) * m;
        } else if (a.gate) {
            const float4 g = *(const float4*)(a.gate + (size_t)n * a.gate_stride + lane * 4);
            const float4 b = *(const float4*)(a.branch + o);
            x.x += g.x * b.x; x.y += g.y * b.y; x.z += g.z * b.z; x.w += g.w * b.w;
        }
        if (a.xout) *(float4*)(a.xout + o) = x;
        if (a.x16) *(uint2*)((unsigned char*)a.x16 + o * 2) = pack4<P>(x.x, x.y, x.z, x.w);
        if (a.x16lo)
            *(uint2*)((unsigned char*)a.x16lo + o * 2) = pack4<P>(x.x - (float)to16<P>(x.x), x.y - (float)to16<P>(x.y),
                                                                 x.z - (float)to16<P>(x.z), x.w - (float)to16<P>(x.w));
        if (a.h16) {
            const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / 256.0f);
            const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
            const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            const float* ad = a.ada + (size_t)n * a.ada_stride + lane * 4;
            const float4 sh = *(const float4*)(ad + a.shift_off), sc = *(const float4*)(ad + a.scale_off);
            const float mm = a.mask_out ? m : 1.0f;
            const float h0 = (d0 * rstd * (1.0f + sc.x) + sh.x) * mm, h1 = (d1 * rstd * (1.0f + sc.y) + sh.y) * mm;
            const float h2 = (d2 * rstd * (1.0f + sc.z) + sh.z) * mm, h3 = (d3 * rstd * (1.0f + sc.w) + sh.w) * mm;
            *(uint2*)((unsigned char*)a.h16 + o * 2) = pack4<P>(h0, h1, h2, h3);
            if (a.h16lo)
                *(uint2*)((unsigned char*)a.h16lo + o * 2) = pack4<P>(h0 - (float)to16<P>(h0), h1 - (float)to16<P>(h1), h2 - (float)to16<P>(h2), h3 - (float)to16<P>(h3));
        }
    }
}

hipError_t launch_train_ln(int dtype, const TrainLnArgs& a, hipStream_t s) {
    int grid = (a.rows + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
    if (grid < 1) grid = 1;
    if (dtype == DT_BF16) hipLaunchKernelGGL((train_ln_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, a);
    else                  hipLaunchKernelGGL((train_ln_kernel<OpF16>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ forward: SiLU (+ dropout, mask)
template <class P>
__global__ __launch_bounds__(256) void silu_drop_kernel(const typename P::elem* a16, typename P::elem* u16, const float* mask,
                                                        int mask_mod, int T, int F, int64_t rows, DropCfg drop) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;       // 8 elements (16 bytes) per thread; F % 8 == 0
    if (i >= rows * F) return;
    const int64_t row = i / F;
    float m = 1.0f;
    if (mask) { const int64_t n = row / T; m = mask[(size_t)(n % mask_mod) * T + (row - n * T)]; }
    const typename P::vec8 a = as_vec8<P>(*(const uint4*)(a16 + i));
    typename P::vec8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        float2 f = make_float2(1.0f, 1.0f);
        if (drop.thresh16) f = drop_factors2(drop, drop_ffn_hash(drop, (unsigned long long)(i + e)));
        o[e] = to16<P>(silu_fast((float)a[e]) * f.x * m);
        o[e + 1] = to16<P>(silu_fast((float)a[e + 1]) * f.y * m);
    }
    *(uint4*)(u16 + i) = __builtin_bit_cast(uint4, o);
}

hipError_t launch_silu_drop(int dtype, const void* a16, void* u16, const float* mask, int mask_mod, int T, int F,
                            int64_t rows, DropCfg drop, hipStream_t s) {
    if (F % 8) return hipErrorInvalidValue;
    const int64_t n4 = rows * F / 8;
    const int grid = (int)((n4 + 255) / 256);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((silu_drop_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, (const __bf16*)a16, (__bf16*)u16, mask, mask_mod, T, F, rows, drop);
    else
        hipLaunchKernelGGL((silu_drop_kernel<OpF16>), dim3(grid), dim3(256), 0, s, (const _Float16*)a16, (_Float16*)u16, mask, mask_mod, T, F, rows, drop);
    return hipGetLastError();
}

template <class P>
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* dU, const typename P::elem* a16, const float* mask,
                                                       int mask_mod, int T, int F, int64_t rows, DropCfg drop,
                                                       typename P::elem* dA16) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;       // 8 elements per thread; F % 8 == 0
    if (i >= rows * F) return;
    const int64_t row = i / F;
    float m = 1.0f;
    if (mask) { const int64_t n = row / T; m = mask[(size_t)(n % mask_mod) * T + (row - n * T)]; }
    const typename P::vec8 a = as_vec8<P>(*(const uint4*)(a16 + i));
    const float4 g0 = *(const float4*)(dU + i), g1 = *(const float4*)(dU + i + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    typename P::vec8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        float2 f = make_float2(1.0f, 1.0f);
        if (drop.thresh16) f = drop_factors2(drop, drop_ffn_hash(drop, (unsigned long long)(i + e)));
        o[e] = to16<P>(g[e] * m * f.x * silu_grad_fast((float)a[e]));
        o[e + 1] = to16<P>(g[e + 1] * m * f.y * silu_grad_fast((float)a[e + 1]));
    }
    *(uint4*)(dA16 + i) = __builtin_bit_cast(uint4, o);
}

hipError_t launch_silu_bwd(int dtype, const float* dU, const void* a16, const float* mask, int mask_mod, int T, int F,
                           int64_t rows, DropCfg drop, void* dA16, hipStream_t s) {
    if (F % 8) return hipErrorInvalidValue;
    const int64_t n4 = rows * F / 8;
    const int grid = (int)((n4 + 255) / 256);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((silu_bwd_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, dU, (const __bf16*)a16, mask, mask_mod, T, F, rows, drop, (__bf16*)dA16);
    else
        hipLaunchKernelGGL((silu_bwd_kernel<OpF16>), dim3(grid), dim3(256), 0, s, dU, (const _Float16*)a16, mask, mask_mod, T, F, rows, drop, (_Float16*)dA16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ backward: per-item channel sums
// Block = one item x one chunk of kRedRows frames, 4 waves; wave w takes frames chunk*kRedRows + w, +4, ...
// K running sums per lane (4 channels each); combined over the 4 waves through LDS; written to part[n][chunk][k][256].
template <int K>
__device__ __forceinline__ void store_parts(float4 (&acc)[K], float* part, int n, int chunk, int chunks) {
    __shared__ float4 red[4][K][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave][k][lane] = acc[k];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 v = red[0][k][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 u = red[w][k][lane];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            *(float4*)(part + (((size_t)n * chunks + chunk) * K + k) * 256 + lane * 4) = v;
        }
    }
}

// max |value| a block saw -> ONE atomic into a group of kMaxCells cells 64 bytes apart (blocks are dealt over the cells: a few
// thousand same-address atomics serialise in the L2 and would set the kernel's time; the consumer takes the maximum of the group).
// Non-negative floats order like their bit patterns; NaN / inf do not set a scale (as absmax_kernel).
__device__ __forceinline__ float absmax_take(float m, float v) { v = fabsf(v); return (v == v && v < 3.0e38f) ? fmaxf(m, v) : m; }
__device__ __forceinline__ void publish_block_max(float m, unsigned* cells) {
    __shared__ float wm_[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wm_[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float b = fmaxf(fmaxf(wm_[0], wm_[1]), fmaxf(wm_[2], wm_[3]));
        if (b > 0.f) atomicMax(cells + (((blockIdx.x + blockIdx.y * 7u) % kMaxCells) << 4), __float_as_uint(b));
    }
}

// Every per-(item, channel) sum of one backward block in ONE launch: site q = {partials of a row kernel, its K sums, where they go,
// the un-scaling pair that was current when the row kernel ran}; block (j, n) reduces flat sum j of item n over the chunks in order.
__global__ __launch_bounds__(256) void reduce_sites_kernel(RedSites S, int chunks) {
    int j = blockIdx.x, q = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) if (t < S.n - 1 && q == t && j >= S.s[t].K) { j -= S.s[t].K; q = t + 1; }
    const RedSite st = S.s[q];
    const int n = blockIdx.y, ch = threadIdx.x;
    float v = 0.f;
    const float* src = st.part + ((size_t)n * chunks * st.K + j) * 256 + ch;
    const size_t cs = (size_t)st.K * 256;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {        // eight chunk partials requested together (the loop was a chain of L2 latencies: 40 us for 16 MB), added in order
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(c + u) * cs];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; c < chunks; ++c) v += src[(size_t)c * cs];
    if (st.unscale) v *= st.unscale[1];
    st.out[(size_t)n * st.out_stride + st.off[j] + ch] = v;
}
hipError_t launch_reduce_sites(const RedSites& S, int n_items, int chunks, hipStream_t s) {
    if (S.n < 1 || S.n > 6) return hipErrorInvalidValue;
    int total = 0;
    for (int q = 0; q < S.n; ++q) { if (S.s[q].K < 1 || S.s[q].K > 2) return hipErrorInvalidValue; total += S.s[q].K; }
    hipLaunchKernelGGL(reduce_sites_kernel, dim3(total, n_items), dim3(256), 0, s, S, chunks);
    return hipGetLastError();
}

struct RedOff { int off[4]; };
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* part, int chunks, int K, float* out, int out_stride,
                                                            RedOff off, int accumulate, const float* unscale) {
    const int k = blockIdx.x, n = blockIdx.y, ch = threadIdx.x;
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += part[(((size_t)n * chunks + c) * K + k) * 256 + ch];
    if (unscale) v *= unscale[1];
    float* dst = out + (size_t)n * out_stride + off.off[k] + ch;
    *dst = accumulate ? *dst + v : v;
}

hipError_t launch_reduce_parts(const float* part, int n_items, int chunks, int K, float* out, int out_stride,
                               const int* out_off, int accumulate, const float* unscale, hipStream_t s) {
    if (K < 1 || K > 4) return hipErrorInvalidValue;
    RedOff off;
    for (int k = 0; k < 4; ++k) off.off[k] = k < K ? out_off[k] : 0;     // out_off is a HOST array
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(K, n_items), dim3(256), 0, s, part, chunks, K, out, out_stride, off, accumulate, unscale);
    return hipGetLastError();
}

// x_out = x_in + gate * branch
template <class P>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* dX, const float* branch, const float* gate, int gate_stride,
                                                       const float* mask, int mask_mod, int T, typename P::elem* dB16,
                                                       float* part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x, n = blockIdx.y, chunks = gridDim.x;
    const float4 g = *(const float4*)(gate + (size_t)n * gate_stride + lane * 4);
    float4 acc[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int t = chunk * kRedRows + wave; t < T && t < (chunk + 1) * kRedRows; t += 4) {
        const size_t o = ((size_t)n * T + t) * 256 + lane * 4;
        const float m = mask ? mask[(size_t)(n % mask_mod) * T + t] : 1.0f;
        const float4 d = *(const float4*)(dX + o);
        const float4 b = *(const float4*)(branch + o);
        acc[0].x += d.x * b.x; acc[0].y += d.y * b.y; acc[0].z += d.z * b.z; acc[0].w += d.w * b.w;
        *(uint2*)(dB16 + o) = pack4<P>(d.x * g.x * m, d.y * g.y * m, d.z * g.z * m, d.w * g.w * m);
    }
    store_parts<1>(acc, part, n, chunk, chunks);
}

hipError_t launch_gate_bwd(int dtype, const float* dX, const float* branch, const float* gate, int gate_stride,
                           const float* mask, int mask_mod, int T, int n_items, void* dB16, float* part, hipStream_t s) {
    dim3 grid(red_chunks(T), n_items);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((gate_bwd_kernel<OpBF16>), grid, dim3(256), 0, s, dX, branch, gate, gate_stride, mask, mask_mod, T, (__bf16*)dB16, part);
    else
        hipLaunchKernelGGL((gate_bwd_kernel<OpF16>), grid, dim3(256), 0, s, dX, branch, gate, gate_stride, mask, mask_mod, T, (_Float16*)dB16, part);
    return hipGetLastError();
}

// h = (LN(x) * (1 + sc) + sh) [* mask]
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* x, const float* dH, const float* ada, int ada_stride,
                                                     int scale_off, const float* mask, int mask_mod, int mask_out, int T,
                                                     float* dX, float* part, const float* dh_scale, unsigned* amax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x, n = blockIdx.y, chunks = gridDim.x;
    const float dhs = dh_scale ? dh_scale[1] : 1.0f;
    float mx = 0.f;
    const float4 sc = *(const float4*)(ada + (size_t)n * ada_stride + scale_off + lane * 4);
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int t = chunk * kRedRows + wave; t < T && t < (chunk + 1) * kRedRows; t += 4) {
        const size_t o = ((size_t)n * T + t) * 256 + lane * 4;
        const float mm = ((mask_out && mask) ? mask[(size_t)(n % mask_mod) * T + t] : 1.0f) * dhs;
        const float4 xv = *(const float4*)(x + o);
        float4 g = *(const float4*)(dH + o);
        g.x *= mm; g.y *= mm; g.z *= mm; g.w *= mm;
        const float mean = wave_sum(xv.x + xv.y + xv.z + xv.w) * (1.0f / 256.0f);
        const float d0 = xv.x - mean, d1 = xv.y - mean, d2 = xv.z - mean, d3 = xv.w - mean;
        const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float n0 = d0 * rstd, n1 = d1 * rstd, n2 = d2 * rstd, n3 = d3 * rstd;
        acc[0].x += g.x * n0; acc[0].y += g.y * n1; acc[0].z += g.z * n2; acc[0].w += g.w * n3;     // d scale
        acc[1].x += g.x; acc[1].y += g.y; acc[1].z += g.z; acc[1].w += g.w;                         // d shift
        const float e0 = g.x * (1.0f + sc.x), e1 = g.y * (1.0f + sc.y), e2 = g.z * (1.0f + sc.z), e3 = g.w * (1.0f + sc.w);
        const float m1 = wave_sum(e0 + e1 + e2 + e3) * (1.0f / 256.0f);
        const float m2 = wave_sum(e0 * n0 + e1 * n1 + e2 * n2 + e3 * n3) * (1.0f / 256.0f);
        float4 d = *(const float4*)(dX + o);
        d.x += rstd * (e0 - m1 - n0 * m2); d.y += rstd * (e1 - m1 - n1 * m2);
        d.z += rstd * (e2 - m1 - n2 * m2); d.w += rstd * (e3 - m1 - n3 * m2);
        *(float4*)(dX + o) = d;
        mx = absmax_take(absmax_take(absmax_take(absmax_take(mx, d.x), d.y), d.z), d.w);
    }
    store_parts<2>(acc, part, n, chunk, chunks);
    if (amax) publish_block_max(mx, amax);      // the re-centring point that follows needs max |dX|: no separate 65-MB pass
}

hipError_t launch_ln_bwd(const float* x, const float* dH, const float* ada, int ada_stride, int scale_off,
                         const float* mask, int mask_mod, int mask_out, int T, int n_items, float* dX, float* part,
                         const float* dh_scale, unsigned* amax, hipStream_t s) {
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(red_chunks(T), n_items), dim3(256), 0, s, x, dH, ada, ada_stride, scale_off,
                       mask, mask_mod, mask_out, T, dX, part, dh_scale, amax);
    return hipGetLastError();
}

// x = (gamma * xpre + beta) * mask
template <class P>
__global__ __launch_bounds__(256) void film_bwd_kernel(const float* xpre, const float* film, int film_stride, int film_mod,
                                                       const float* mask, int mask_mod, int T, float* dX,
                                                       typename P::elem* dX16, float* part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x, n = blockIdx.y, chunks = gridDim.x;
    const float4 ga = *(const float4*)(film + (size_t)(n % film_mod) * film_stride + lane * 4);
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int t = chunk * kRedRows + wave; t < T && t < (chunk + 1) * kRedRows; t += 4) {
        const size_t o = ((size_t)n * T + t) * 256 + lane * 4;
        const float m = mask ? mask[(size_t)(n % mask_mod) * T + t] : 1.0f;
        float4 d = *(const float4*)(dX + o);
        d.x *= m; d.y *= m; d.z *= m; d.w *= m;
        const float4 xp = *(const float4*)(xpre + o);
        acc[0].x += d.x * xp.x; acc[0].y += d.y * xp.y; acc[0].z += d.z * xp.z; acc[0].w += d.w * xp.w;   // d gamma
        acc[1].x += d.x; acc[1].y += d.y; acc[1].z += d.z; acc[1].w += d.w;                               // d beta
        d.x *= ga.x; d.y *= ga.y; d.z *= ga.z; d.w *= ga.w;
        *(float4*)(dX + o) = d;
        if (dX16) *(uint2*)(dX16 + o) = pack4<P>(d.x, d.y, d.z, d.w);
    }
    store_parts<2>(acc, part, n, chunk, chunks);
}

hipError_t launch_film_bwd(int dtype, const float* xpre, const float* film, int film_stride, int film_mod,
                           const float* mask, int mask_mod, int T, int n_items, float* dX, void* dX16, float* part,
                           hipStream_t s) {
    dim3 grid(red_chunks(T), n_items);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((film_bwd_kernel<OpBF16>), grid, dim3(256), 0, s, xpre, film, film_stride, film_mod, mask, mask_mod, T, dX, (__bf16*)dX16, part);
    else
        hipLaunchKernelGGL((film_bwd_kernel<OpF16>), grid, dim3(256), 0, s, xpre, film, film_stride, film_mod, mask, mask_mod, T, dX, (_Float16*)dX16, part);
    return hipGetLastError();
}

template <class P>
__global__ __launch_bounds__(256) void cast16_kernel(const float* x, const float* mask, int mask_mod, int T, int C, int64_t rows,
                                                     const float* scale, typename P::elem* y16) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * C) return;
    float m = scale ? scale[0] : 1.0f;
    if (mask) { const int64_t row = i / C, n = row / T; m *= mask[(size_t)(n % mask_mod) * T + (row - n * T)]; }
    const float4 v = *(const float4*)(x + i);
    *(uint2*)(y16 + i) = pack4<P>(v.x * m, v.y * m, v.z * m, v.w * m);
}

hipError_t launch_cast16(int dtype, const float* x, const float* mask, int mask_mod, int T, int C, int64_t rows,
                         const float* scale, void* y16, hipStream_t s) {
    const int64_t n4 = rows * C / 4;
    const int grid = (int)((n4 + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((cast16_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, x, mask, mask_mod, T, C, rows, scale, (__bf16*)y16);
    else                  hipLaunchKernelGGL((cast16_kernel<OpF16>), dim3(grid), dim3(256), 0, s, x, mask, mask_mod, T, C, rows, scale, (_Float16*)y16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ gradient scaling
// 16-bit gradient operands need the dynamic range of the incoming gradient placed inside the operand type's: f16 has
// 5 exponent bits (normal >= 6e-5) while d loss / d out is ~1 / (sum(mask) * n_feats) per element.  The whole backward
// pass is linear in d loss / d out, so it is multiplied by a power of two on entry (max |g| -> ~2^8) and every fp32
// result is multiplied by the inverse where it leaves the pass.  sc[0] = scale, sc[1] = 1 / scale (on the device: no
// host synchronisation).
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, int64_t n, unsigned* out_bits) {
    float m = 0.f;
    auto take = [&](float v) { v = fabsf(v); if (v == v && v < 3.0e38f) m = fmaxf(m, v); };      // NaN / inf do not set the scale
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-byte loads in flight per lane
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(x + 4 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < 4; ++u) { take(v[u].x); take(v[u].y); take(v[u].z); take(v[u].w); }
    }
    for (; i < n4; i += stride) { const float4 v = *(const float4*)(x + 4 * i); take(v.x); take(v.y); take(v.z); take(v.w); }
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) take(x[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    // ONE atomic per block (thousands of same-address atomics serialise in the L2: they, not the 65 MB read, set this kernel's time)
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));   // non-negative floats order like their bits
}
__global__ void grad_scale_kernel(const unsigned* bits, float* sc) {
    const float m = __uint_as_float(*bits);
    float s = 1.0f;
    if (m > 0.f) {
        int ex = 5 - (int)floorf(log2f(m));           // max |g| * scale in [2^5, 2^6): eleven binades below f16's 65504 for what the
                                                      // kernels up to the next re-centring point add, twenty above its normal minimum
        ex = ex < -60 ? -60 : (ex > 60 ? 60 : ex);
        s = exp2f((float)ex);
    }
    sc[0] = s; sc[1] = 1.0f / s;
}
hipError_t launch_grad_scale(const float* g, int64_t n, unsigned* bits, float* sc, hipStream_t s) {
    hipError_t e = hipMemsetAsync(bits, 0, 4, s);
    if (e != hipSuccess) return e;
    int grid = (int)((n / 4 + 255) / 256); if (grid > 512) grid = 512; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, g, n, bits);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, s, bits, sc);
    return hipGetLastError();
}
__global__ void qkv_scales_kernel(const unsigned* bits3, const float* gsc, float* qs) {
    float f[3], mx = 0.f;
    for (int i = 0; i < 3; ++i) {
        const float m = __uint_as_float(bits3[i]);
        mx = fmaxf(mx, m);
        int ex = m > 0.f ? 8 - (int)floorf(log2f(m)) : 0;
        ex = ex < -60 ? -60 : (ex > 60 ? 60 : ex);
        f[i] = exp2f((float)ex);
    }
    int exc = mx > 0.f ? 8 - (int)floorf(log2f(mx)) : 0;
    exc = exc < -60 ? -60 : (exc > 60 ? 60 : exc);
    const float fc = exp2f((float)exc);
    qs[0] = fc; qs[1] = 1.0f / fc;
    for (int i = 0; i < 3; ++i) { qs[2 + 2 * i] = f[i] * gsc[0]; qs[3 + 2 * i] = 1.0f / (f[i] * gsc[0]); qs[8 + i] = f[i]; }
}
hipError_t launch_qkv_grad_scales(const float* dq, const float* dk, const float* dv, int64_t n, const float* gsc,
                                  unsigned* bits3, float* qs, hipStream_t s, bool have_max) {
    if (!have_max) {
        hipError_t e = hipMemsetAsync(bits3, 0, 12, s);
        if (e != hipSuccess) return e;
        int grid = (int)((n / 4 + 255) / 256); if (grid > 512) grid = 512; if (grid < 1) grid = 1;
        hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, dq, n, bits3);
        hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, dk, n, bits3 + 1);
        hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, dv, n, bits3 + 2);
    }
    hipLaunchKernelGGL(qkv_scales_kernel, dim3(1), dim3(1), 0, s, bits3, gsc, qs);
    return hipGetLastError();
}
__global__ void grad_rescale_kernel(const unsigned* bits, float* sc) {
    const float m = __uint_as_float(*bits);
    float f = 1.0f;
    if (m > 0.f && (m < 2.0f || m >= 512.0f)) {
        int ex = 5 - (int)floorf(log2f(m));
        const int cur = (int)floorf(log2f(sc[0]));
        if (cur + ex > 100) ex = 100 - cur;              // keep the total scale a finite fp32 power of two
        if (cur + ex < -100) ex = -100 - cur;
        f = exp2f((float)ex);
    }
    sc[2] = f;
    sc[0] *= f; sc[1] = 1.0f / sc[0];
}
hipError_t launch_grad_rescale(const float* g, int64_t n, unsigned* bits, float* sc, hipStream_t s) {
    hipError_t e = hipMemsetAsync(bits, 0, 4, s);
    if (e != hipSuccess) return e;
    int grid = (int)((n / 4 + 255) / 256); if (grid > 512) grid = 512; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, g, n, bits);
    hipLaunchKernelGGL(grad_rescale_kernel, dim3(1), dim3(1), 0, s, bits, sc);
    return hipGetLastError();
}
// max |x| into a cell GROUP (publish_block_max): the fall-back of a re-centring point whose producer is a GEMM epilogue
__global__ __launch_bounds__(256) void absmax_cells_kernel(const float* x, int64_t n, unsigned* cells) {
    float m = 0.f;
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(x + 4 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < 4; ++u) m = absmax_take(absmax_take(absmax_take(absmax_take(m, v[u].x), v[u].y), v[u].z), v[u].w);
    }
    for (; i < n4; i += stride) { const float4 v = *(const float4*)(x + 4 * i); m = absmax_take(absmax_take(absmax_take(absmax_take(m, v.x), v.y), v.z), v.w); }
    publish_block_max(m, cells);
}
// One launch per re-centring point: f from the published maximum and the CURRENT pair (both read-only here), the tensor multiplied
// by f (nothing to do when f == 1: the usual case), the NEXT pair {s f, 1 / (s f), f} written to its own slot by one thread --
// later kernels are handed the new slot, so nothing races with the blocks still reading the old one.
__global__ __launch_bounds__(256) void rescale_apply_kernel(float* a, int64_t n4, const unsigned* cells, const float* sc, float* sc_next) {
    unsigned mb = 0u;
#pragma unroll
    for (int c = 0; c < kMaxCells; ++c) { const unsigned v = cells[c << 4]; mb = v > mb ? v : mb; }
    const float m = __uint_as_float(mb), s0 = sc[0];
    float f = 1.0f;
    if (m > 0.f && (m < 2.0f || m >= 512.0f)) {
        int ex = 5 - (int)floorf(log2f(m));
        const int cur = (int)floorf(log2f(s0));
        if (cur + ex > 100) ex = 100 - cur;              // keep the total scale a finite fp32 power of two
        if (cur + ex < -100) ex = -100 - cur;
        f = exp2f((float)ex);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc_next[0] = s0 * f; sc_next[1] = 1.0f / (s0 * f); sc_next[2] = f; }
    if (f == 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = *(float4*)(a + 4 * i);
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        *(float4*)(a + 4 * i) = v;
    }
}
hipError_t launch_recentre(float* a, int64_t n, unsigned* cells, bool have_max, const float* sc, float* sc_next, hipStream_t s) {
    if (n & 3) return hipErrorInvalidValue;
    if (!have_max) {
        int grid = (int)((n / 4 + 255) / 256); if (grid > 512) grid = 512; if (grid < 1) grid = 1;
        hipLaunchKernelGGL(absmax_cells_kernel, dim3(grid), dim3(256), 0, s, a, n, cells);
    }
    int grid = (int)((n / 4 + 255) / 256); if (grid > 512) grid = 512; if (grid < 1) grid = 1;      // (usually f == 1: every block reads the cells and leaves)
    hipLaunchKernelGGL(rescale_apply_kernel, dim3(grid), dim3(256), 0, s, a, n / 4, cells, sc, sc_next);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void scale_by_kernel(float* a, int64_t n4, const float* sc) {
    const float f = sc[2];
    if (f == 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = *(float4*)(a + 4 * i);
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        *(float4*)(a + 4 * i) = v;
    }
}
hipError_t launch_scale_by(float* a, int64_t n, const float* sc, hipStream_t s) {
    if (n & 3) return hipErrorInvalidValue;
    int grid = (int)((n / 4 + 255) / 256); if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(scale_by_kernel, dim3(grid), dim3(256), 0, s, a, n / 4, sc);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void scale_inplace_kernel(float* a, int64_t n, const float* sc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] *= sc[1];
}
hipError_t launch_unscale_inplace(float* a, int64_t n, const float* sc, hipStream_t s) {
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n, sc);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* a, const float* b, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 x = *(const float4*)(a + i);
    const float4 y = *(const float4*)(b + i);
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    *(float4*)(a + i) = x;
}
__global__ __launch_bounds__(256) void add_rescaled_kernel(float* a, const float* b, int64_t n, const float* sc, const float* sc_b, unsigned* amax) {
    float mx = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const float f = sc[0] / sc_b[0];
        float4 x = *(float4*)(a + i);
        const float4 y = *(const float4*)(b + i);
        x.x += y.x * f; x.y += y.y * f; x.z += y.z * f; x.w += y.w * f;
        *(float4*)(a + i) = x;
        mx = absmax_take(absmax_take(absmax_take(absmax_take(mx, x.x), x.y), x.z), x.w);
    }
    if (amax) publish_block_max(mx, amax);      // (every thread of the block reaches this: the loop has no early return)
}
hipError_t launch_add_rescaled(float* a, const float* b, int64_t n, const float* sc, const float* sc_b, unsigned* amax, hipStream_t s) {
    if (n & 3) return hipErrorInvalidValue;
    int grid = (int)((n / 4 + 255) / 256); if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(add_rescaled_kernel, dim3(grid), dim3(256), 0, s, a, b, n, sc, sc_b, amax);
    return hipGetLastError();
}
__global__ void copy_scalars_kernel(float* dst, const float* src, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x]; }
hipError_t launch_copy_scalars(float* dst, const float* src, int n, hipStream_t s) {
    hipLaunchKernelGGL(copy_scalars_kernel, dim3(1), dim3(64), 0, s, dst, src, n);
    return hipGetLastError();
}
hipError_t launch_add_inplace(float* a, const float* b, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3((int)((n / 4 + 255) / 256)), dim3(256), 0, s, a, b, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ wgrad operand transposes
// XT[s][j*Cin + ci][rl] = X[n][t + j - taps/2][ci] for the global row r = s*Rs + rl = n*T + t (zero when r >= N*T or
// the shifted frame leaves the item).  Block: 64 rows x 64 channels, through LDS (66 rows with the halo).
template <class P>
__global__ __launch_bounds__(256) void wgrad_xt_kernel(const typename P::elem* x0, int c0, const typename P::elem* x1, int c1,
                                                       int n_items, int T, int taps, int Rs, typename P::elem* xt) {
    __shared__ typename P::elem tile[66][64 + 2];
    const int cin = c0 + c1;
    const int64_t R = (int64_t)n_items * T;
    const int64_t r0 = (int64_t)blockIdx.x * 64;        // first padded-global row of the block: (s, rl0)
    const int sidx = (int)(r0 / Rs), rl0 = (int)(r0 % Rs);
    const int64_t g0 = (int64_t)sidx * Rs + rl0;         // == r0: rows are numbered contiguously across splits
    const int ch0 = blockIdx.y * 64;
    const typename P::elem* src; int cs, coff;
    if (ch0 < c0) { src = x0; cs = c0; coff = ch0; } else { src = x1; cs = c1; coff = ch0 - c0; }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 channels x 4 rows per pass
    for (int rr = ty; rr < 66; rr += 4) {
        const int64_t r = g0 + rr - 1;                           // LDS row rr holds global row g0 + rr - 1
        typename P::elem v = (typename P::elem)0.0f;
        if (r >= 0 && r < R) v = src[(size_t)r * cs + coff + tx];
        tile[rr][tx] = v;
    }
    __syncthreads();
    // output: for tap j, channel c (64), column rl (64): value = X[row g0 + rl + j - taps/2] if same item & valid
    const int half = taps / 2;
    for (int j = 0; j < taps; ++j)
        for (int c = ty; c < 64; c += 4) {
            const int rl = tx;
            const int64_t r = g0 + rl;                           // the row this column stands for (n, t)
            typename P::elem v = (typename P::elem)0.0f;
            if (r < R) {
                const int t = (int)(r % T) + j - half;
                if (t >= 0 && t < T) v = tile[rl + 1 + j - half][c];
            }
            xt[((size_t)sidx * taps * cin + (size_t)j * cin + ch0 + c) * Rs + rl0 + rl] = v;
        }
}

hipError_t launch_wgrad_xt(int dtype, const void* x0, int c0, const void* x1, int c1, int n_items, int T, int taps,
                           int S, int Rs, void* xt, hipStream_t s) {
    if ((c0 & 63) || (c1 & 63) || (Rs & 63)) return hipErrorInvalidValue;
    dim3 grid((unsigned)((int64_t)S * Rs / 64), (unsigned)((c0 + c1) / 64));
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((wgrad_xt_kernel<OpBF16>), grid, dim3(256), 0, s, (const __bf16*)x0, c0, (const __bf16*)x1, c1, n_items, T, taps, Rs, (__bf16*)xt);
    else
        hipLaunchKernelGGL((wgrad_xt_kernel<OpF16>), grid, dim3(256), 0, s, (const _Float16*)x0, c0, (const _Float16*)x1, c1, n_items, T, taps, Rs, (_Float16*)xt);
    return hipGetLastError();
}

// dYT[s][co][rl] = dY[r = s*Rs + rl][co] (zero for r >= R);  part_b[rowblock][co] = sum over the block's 64 rows
template <class P>
__global__ __launch_bounds__(256) void wgrad_dyt_kernel(const typename P::elem* dy, int cout, int64_t R, int Rs,
                                                        typename P::elem* dyt, float* part_b) {
    __shared__ typename P::elem tile[64][64 + 2];
    __shared__ float colsum[4][64];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int sidx = (int)(r0 / Rs), rl0 = (int)(r0 % Rs);
    const int ch0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float cs = 0.f;
    for (int rr = ty; rr < 64; rr += 4) {
        const int64_t r = r0 + rr;
        typename P::elem v = (typename P::elem)0.0f;
        if (r < R) v = dy[(size_t)r * cout + ch0 + tx];
        tile[rr][tx] = v;
        cs += (float)v;
    }
    colsum[ty][tx] = cs;
    __syncthreads();
    if (ty == 0 && part_b) part_b[(size_t)blockIdx.x * cout + ch0 + tx] = colsum[0][tx] + colsum[1][tx] + colsum[2][tx] + colsum[3][tx];
    for (int c = ty; c < 64; c += 4)
        dyt[((size_t)sidx * cout + ch0 + c) * Rs + rl0 + tx] = tile[tx][c];
}

hipError_t launch_wgrad_dyt(int dtype, const void* dy, int cout, int64_t R, int S, int Rs, void* dyt, float* part_b,
                            hipStream_t s) {
    if ((cout & 63) || (Rs & 63)) return hipErrorInvalidValue;
    dim3 grid((unsigned)((int64_t)S * Rs / 64), (unsigned)(cout / 64));
    if (dtype == DT_BF16) hipLaunchKernelGGL((wgrad_dyt_kernel<OpBF16>), grid, dim3(256), 0, s, (const __bf16*)dy, cout, R, Rs, (__bf16*)dyt, part_b);
    else                  hipLaunchKernelGGL((wgrad_dyt_kernel<OpF16>), grid, dim3(256), 0, s, (const _Float16*)dy, cout, R, Rs, (_Float16*)dyt, part_b);
    return hipGetLastError();
}

// dW[co - co_start][ci_off + ci][j] = sum_s partial[s][j*cin + ci][co]  for co in [co_start, co_start + co_cnt), ci < ci_cnt
// (partial: [S][taps*cin][cout] fp32; dW: reference layout (co_cnt, cin_total, taps))
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, int S, int cin, int cout, int taps, float* dW,
                                                           int cin_total, int ci_off, int ci_cnt, int co_start, int co_cnt,
                                                           const float* unscale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over (j*cin + ci) x co, co fastest
    const int64_t total = (int64_t)taps * cin * cout;
    if (idx >= total) return;
    const int co = (int)(idx % cout);
    const int jc = (int)(idx / cout);
    const int j = jc / cin, ci = jc % cin;
    if (ci >= ci_cnt || co < co_start || co >= co_start + co_cnt) return;
    float v = 0.f;
    for (int sI = 0; sI < S; ++sI) v += partial[(size_t)sI * total + idx];
    if (unscale) v *= unscale[1];
    dW[((size_t)(co - co_start) * cin_total + ci_off + ci) * taps + j] = v;
}

hipError_t launch_wgrad_reduce(const float* partial, int S, int cin, int cout, int taps, float* dW, int cin_total,
                               int ci_off, int ci_cnt, int co_start, int co_cnt, const float* unscale, hipStream_t s) {
    const int64_t total = (int64_t)taps * cin * cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, partial, S, cin, cout, taps,
                       dW, cin_total, ci_off, ci_cnt, co_start, co_cnt, unscale);
    return hipGetLastError();
}

// The same reduction for EVERY output of one weight-gradient GEMM in one launch (the fused q/k/v projection has three row blocks with
// their own un-scaling pairs), plus the bias gradients from the [S][cout] column-sum planes wgrad_tn_kernel writes beside its tiles:
// blocks [0, nb_w) reduce dW elements (co fastest), the last ceil(cout / 256) blocks one bias channel per thread.  Plane order fixed.
struct WgradRed3 { WgradRed o[3]; int n; };
// Tiled form (round 6): block = 64 output channels x 64 input channels x all taps.  The planes are read as float4 rows along co (the
// partial tiles' fast axis), four planes in flight, summed IN PLANE ORDER (bit-identical to the element-per-thread form it replaces);
// the sums go through an LDS tile [co][ci * taps + j] and out as 256-byte runs of the reference layout (co, ci, j) -- the old form
// wrote every element as a lone 4-byte store 3 KB away from its neighbour's (786 k partial-line writes for an FFN-sized gradient).
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const float* partial, const float* part_b, int S, int cin, int cout,
                                                                 WgradRed3 R, unsigned nb_w, int tiles_ci) {
    constexpr int ROW = 64 * TAPS + 1;
    __shared__ float tile[64 * ROW];
    const int64_t total = (int64_t)TAPS * cin * cout;
    if (blockIdx.x >= nb_w) {
        const int co = (int)(blockIdx.x - nb_w) * 256 + threadIdx.x;
        if (co >= cout || !part_b) return;
        int k = -1;
#pragma unroll
        for (int q = 0; q < 3; ++q) if (q < R.n && R.o[q].db && co >= R.o[q].co_start && co < R.o[q].co_start + R.o[q].co_cnt) k = q;
        if (k < 0) return;
        float v = 0.f;
        for (int sI = 0; sI < S; ++sI) v += part_b[(size_t)sI * cout + co];
        if (R.o[k].unscale) v *= R.o[k].unscale[1];
        R.o[k].db[co - R.o[k].co_start] = v;
        return;
    }
    const int co0 = (int)(blockIdx.x / tiles_ci) * 64, ci0 = (int)(blockIdx.x % tiles_ci) * 64;
    int k = -1;      // the output this tile's 64 channels belong to (row blocks of the fused q/k/v projection are multiples of 256)
#pragma unroll
    for (int q = 0; q < 3; ++q) if (q < R.n && R.o[q].dW && co0 >= R.o[q].co_start && co0 < R.o[q].co_start + R.o[q].co_cnt) k = q;
    if (k < 0) return;
    const WgradRed o = R.o[k];
    if (ci0 >= o.ci_cnt) return;
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
    for (int j = 0; j < TAPS; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cil = rg + 16 * u, ci = ci0 + cil;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < cin) {
                const float* p = partial + ((size_t)j * cin + ci) * cout + co0 + cq * 4;
                int sI = 0;
                for (; sI + 4 <= S; sI += 4) {          // four planes requested together, added in plane order
                    const float4 v0 = *(const float4*)(p + (size_t)(sI + 0) * total), v1 = *(const float4*)(p + (size_t)(sI + 1) * total);
                    const float4 v2 = *(const float4*)(p + (size_t)(sI + 2) * total), v3 = *(const float4*)(p + (size_t)(sI + 3) * total);
                    acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
                    acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
                    acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
                    acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
                }
                for (; sI < S; ++sI) { const float4 v = *(const float4*)(p + (size_t)sI * total); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            }
            float* t = tile + (cq * 4) * ROW + cil * TAPS + j;
            t[0] = acc.x; t[ROW] = acc.y; t[2 * ROW] = acc.z; t[3 * ROW] = acc.w;
        }
    __syncthreads();
    const float us = o.unscale ? o.unscale[1] : 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nci = min(64, o.ci_cnt - ci0);           // valid input channels of this tile
    for (int col = wave; col < 64; col += 4) {
        float* dst = o.dW + ((size_t)(co0 + col - o.co_start) * o.cin_total + o.ci_off + ci0) * TAPS;
        for (int e = lane; e < nci * TAPS; e += 64) dst[e] = tile[col * ROW + e] * us;
    }
}

hipError_t launch_wgrad_reduce_multi(const float* partial, const float* part_b, int S, int cin, int cout, int taps, const WgradRed* outs,
                                     int n_outs, hipStream_t s) {
    if (n_outs < 1 || n_outs > 3 || (cout & 63) || (taps != 1 && taps != 3)) return hipErrorInvalidValue;
    WgradRed3 R; memset(&R, 0, sizeof(R));
    R.n = n_outs;
    bool need_b = false;
    for (int k = 0; k < n_outs; ++k) { R.o[k] = outs[k]; need_b = need_b || outs[k].db; if (outs[k].co_start & 63) return hipErrorInvalidValue; }
    if (need_b && !part_b) return hipErrorInvalidValue;
    const int tiles_ci = (cin + 63) / 64;
    const unsigned nb_w = (unsigned)((cout / 64) * tiles_ci), nb_b = need_b ? (unsigned)((cout + 255) / 256) : 0u;
    if (taps == 3) hipLaunchKernelGGL(wgrad_reduce_multi_kernel<3>, dim3(nb_w + nb_b), dim3(256), 0, s, partial, part_b, S, cin, cout, R, nb_w, tiles_ci);
    else           hipLaunchKernelGGL(wgrad_reduce_multi_kernel<1>, dim3(nb_w + nb_b), dim3(256), 0, s, partial, part_b, S, cin, cout, R, nb_w, tiles_ci);
    return hipGetLastError();
}

// block = 16 channels x 64 row-block groups (cout / 16 blocks: the earlier 64 x 16 shape ran a 256-channel bias on 4 blocks);
// fixed summation order (group partials combined 0..63): deterministic
__global__ __launch_bounds__(1024) void bias_reduce_kernel(const float* part_b, int rowblocks, int cout, float* db, int co_start,
                                                           int co_cnt, const float* unscale) {
    __shared__ float red[64][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int co = blockIdx.x * 16 + tx;
    float v = 0.f;
    if (co < cout)
        for (int rb = ty; rb < rowblocks; rb += 64) v += part_b[(size_t)rb * cout + co];
    red[ty][tx] = v;
    __syncthreads();
    if (ty == 0 && co < cout && co >= co_start && co < co_start + co_cnt) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 64; ++g) t += red[g][tx];
        if (unscale) t *= unscale[1];
        db[co - co_start] = t;
    }
}

hipError_t launch_bias_reduce(const float* part_b, int rowblocks, int cout, float* db, int co_start, int co_cnt,
                              const float* unscale, hipStream_t s) {
    hipLaunchKernelGGL(bias_reduce_kernel, dim3((cout + 15) / 16), dim3(1024), 0, s, part_b, rowblocks, cout, db, co_start, co_cnt, unscale);
    return hipGetLastError();
}

// dgrad weights: Wd[ci][j][col_off + co] = W[co][ci_off + ci][taps - 1 - j]; rows ci >= ci_cnt and columns co >= cout stay zero
template <class P>
__global__ void pack_weight_t_kernel(const float* src, int cout, int cin_total, int taps, int ci_off, int ci_cnt,
                                     typename P::elem* dst, int ld, int col_off) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)ci_cnt * taps * cout;
    if (idx >= total) return;
    const int co = (int)(idx % cout);
    const int j = (int)((idx / cout) % taps);
    const int ci = (int)(idx / ((size_t)cout * taps));
    const float v = src[((size_t)co * cin_total + ci_off + ci) * taps + (taps - 1 - j)];
    dst[((size_t)ci * taps + j) * ld + col_off + co] = to16<P>(v);
}

hipError_t launch_pack_weight_t(int dtype, const float* src, int cout, int cin_total, int taps, int ci_off, int ci_cnt,
                                void* dst, int cin_p, int ld, int col_off, hipStream_t s) {
    (void)cin_p;
    const size_t total = (size_t)ci_cnt * taps * cout;
    const int grid = (int)((total + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((pack_weight_t_kernel<OpBF16>), dim3(grid), dim3(256), 0, s, src, cout, cin_total, taps, ci_off, ci_cnt, (__bf16*)dst, ld, col_off);
    else                  hipLaunchKernelGGL((pack_weight_t_kernel<OpF16>), dim3(grid), dim3(256), 0, s, src, cout, cin_total, taps, ci_off, ci_cnt, (_Float16*)dst, ld, col_off);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ merged re-pack (launch.h: PackJob)
template <class P>
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                 // block-uniform binary search: the job whose block range holds blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    const size_t idx = (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x;
    if (j.kind == 0) {              // pack_weight_kernel (misc_kernels.hip)
        const size_t total = (size_t)j.cout * j.K * j.slice_w;
        if (idx >= total) return;
        const int ci = (int)(idx % j.slice_w);
        const int t = (int)((idx / j.slice_w) % j.K);
        const int co = (int)(idx / ((size_t)j.slice_w * j.K));
        float v = 0.f;
        if (ci < j.ci_cnt) v = j.src[((size_t)co * j.cin_total + j.ci_off + ci) * j.K + t];
        if (j.lo) v -= (float)to16<P>(v);
        ((typename P::elem*)j.dst)[((size_t)(j.row_off + co) * j.K + t) * j.cin_p + j.col_off + ci] = to16<P>(v);
    } else if (j.kind == 1) {       // pack_weight_t_kernel (above); cin_p carries its `ld`
        const size_t total = (size_t)j.ci_cnt * j.K * j.cout;
        if (idx >= total) return;
        const int co = (int)(idx % j.cout);
        const int t = (int)((idx / j.cout) % j.K);
        const int ci = (int)(idx / ((size_t)j.cout * j.K));
        const float v = j.src[((size_t)co * j.cin_total + j.ci_off + ci) * j.K + (j.K - 1 - t)];
        ((typename P::elem*)j.dst)[((size_t)ci * j.K + t) * j.cin_p + j.col_off + co] = to16<P>(v);
    } else if (j.kind == 3) {       // pack_ffn_stream_kernel (misc_kernels.hip): lo = stage, cout = F
        if (idx >= (size_t)j.cout * 256 * 3) return;
        size_t so, dof;
        ffn_stream_index(idx, j.lo, j.cout, &so, &dof);
        ((typename P::elem*)j.dst)[dof] = to16<P>(j.src[so]);
    } else if (j.kind == 5) {       // pack_ffn_wino_kernel (misc_kernels.hip): lo = stage, cout = F; f16 engines only
        if (idx >= (size_t)j.cout * 256 * 3) return;
        size_t so, dof; int pl;
        ffn_wino_index(idx, j.lo, j.cout, &so, &dof, &pl);
        ((typename P::elem*)j.dst)[dof] = to16<P>(ffn_wino_plane(j.src + so, pl));
    } else if (j.kind == 4) {       // pack_qkv_frag_kernel (misc_kernels.hip): row_off = 256 * plane
        if (idx >= 256 * 256) return;
        ((typename P::elem*)j.dst)[qkv_frag_index(j.row_off >> 8, (int)(idx >> 8), (int)(idx & 255))] = to16<P>(j.src[idx]);
    } else {                        // fp32 copy (biases)
        if (idx < (size_t)j.cout) ((float*)j.dst)[idx] = j.src[idx];
    }
}

hipError_t launch_pack_jobs(int dtype, const PackJob* jobs_dev, int njobs, unsigned nblocks, hipStream_t s) {
    if (njobs < 1 || nblocks < 1) return hipSuccess;
    if (dtype == DT_BF16) hipLaunchKernelGGL((pack_jobs_kernel<OpBF16>), dim3(nblocks), dim3(256), 0, s, jobs_dev, njobs);
    else                  hipLaunchKernelGGL((pack_jobs_kernel<OpF16>), dim3(nblocks), dim3(256), 0, s, jobs_dev, njobs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ small fp32 linears: backward
// (adaLN / FiLM / time-MLP linears on per-item vectors: n = batch items, a few hundred inputs and outputs.  Register-blocked so
//  that every loaded value feeds 8 multiply-adds -- one output per thread cost 2 loads and, with SiLU on the input, one expf per
//  multiply-add: 35 us per launch, 27 launches per step.)
// dW[o][k] += sum_n dout[n][o] * act(in[n][k]) ; db[o] += sum_n dout[n][o].  block = 64 inputs x 8 consecutive outputs, its 4
// waves take every fourth item (the loop is a chain of L2-latency loads: four short chains + a fixed-order LDS combine)
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* in, const float* dout, int n, int k, int o, int silu_in,
                                                           float* dW, float* db, int accumulate) {
    __shared__ float red[4][16][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int ki = blockIdx.x * 64 + tx, o0 = blockIdx.y * 8;
    float acc[8], accb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j] = 0.f; accb[j] = 0.f; }
    const bool full = o0 + 8 <= o && (o & 3) == 0;
    if (ki < k) {
#pragma unroll 4
        for (int ni = g; ni < n; ni += 4) {
            float x = in[(size_t)ni * k + ki];
            if (silu_in) x = silu_f(x);
            float d[8];
            if (full) {
                const float4 d0 = *(const float4*)(dout + (size_t)ni * o + o0), d1 = *(const float4*)(dout + (size_t)ni * o + o0 + 4);
                d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = o0 + j < o ? dout[(size_t)ni * o + o0 + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] += d[j] * x; accb[j] += d[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[g][j][tx] = acc[j]; red[g][8 + j][tx] = accb[j]; }
    __syncthreads();
    if (ki >= k) return;
    for (int j = g; j < 8; j += 4) {          // wave g finishes outputs o0 + g, o0 + g + 4
        if (o0 + j >= o) continue;
        const float t = ((red[0][j][tx] + red[1][j][tx]) + red[2][j][tx]) + red[3][j][tx];
        float* w = dW + (size_t)(o0 + j) * k + ki;
        *w = accumulate ? *w + t : t;
        if (ki == 0 && db) {
            const float tb = ((red[0][8 + j][0] + red[1][8 + j][0]) + red[2][8 + j][0]) + red[3][8 + j][0];
            db[o0 + j] = accumulate ? db[o0 + j] + tb : tb;
        }
    }
}

// Several layers of one shape in ONE launch (grid.z = layer): dW_j / db_j written (not accumulated).
__global__ __launch_bounds__(256) void linear_bwd_w_multi_kernel(LinBwdJobs J, int n, int k, int o, int silu_in) {
    __shared__ float red[4][16][64];
    const float* in = J.in[blockIdx.z]; const float* dout = J.dout[blockIdx.z];
    float* dW = J.dW[blockIdx.z]; float* db = J.db[blockIdx.z];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int ki = blockIdx.x * 64 + tx, o0 = blockIdx.y * 8;
    float acc[8], accb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j] = 0.f; accb[j] = 0.f; }
    const bool full = o0 + 8 <= o && (o & 3) == 0;
    if (ki < k) {
#pragma unroll 4
        for (int ni = g; ni < n; ni += 4) {
            float x = in[(size_t)ni * k + ki];
            if (silu_in) x = silu_f(x);
            float d[8];
            if (full) {
                const float4 d0 = *(const float4*)(dout + (size_t)ni * o + o0), d1 = *(const float4*)(dout + (size_t)ni * o + o0 + 4);
                d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = o0 + j < o ? dout[(size_t)ni * o + o0 + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] += d[j] * x; accb[j] += d[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[g][j][tx] = acc[j]; red[g][8 + j][tx] = accb[j]; }
    __syncthreads();
    if (ki >= k) return;
    for (int j = g; j < 8; j += 4) {
        if (o0 + j >= o) continue;
        dW[(size_t)(o0 + j) * k + ki] = ((red[0][j][tx] + red[1][j][tx]) + red[2][j][tx]) + red[3][j][tx];
        if (ki == 0 && db) db[o0 + j] = ((red[0][8 + j][0] + red[1][8 + j][0]) + red[2][8 + j][0]) + red[3][8 + j][0];
    }
}
hipError_t launch_linear_bwd_w_multi(const LinBwdJobs& J, int n, int k, int o, int silu_in, hipStream_t s) {
    if (J.n < 1 || J.n > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(linear_bwd_w_multi_kernel, dim3((unsigned)((k + 63) / 64), (unsigned)((o + 7) / 8), J.n), dim3(256), 0, s, J, n, k, o, silu_in);
    return hipGetLastError();
}
// din[n][k] (+)= act'(in[n][k]) * sum_j sum_o dout_j[n][o] * W_j[o][k] over the J.n layers (they share the input `in`): one launch,
// layers summed in order inside each thread group (deterministic)
__global__ __launch_bounds__(1024) void linear_bwd_in_multi_kernel(LinBwdJobs J, const float* in, int n, int k, int o, int silu_in, float* din, int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ni = blockIdx.y, ki = blockIdx.x * 64 + tx;
    float acc = 0.f;
    if (ki < k) {
        for (int j = 0; j < J.n; ++j) {
            const float* dout = J.dout[j]; const float* W = J.W[j];
#pragma unroll 8
            for (int oi = ty; oi < o; oi += 16) acc += dout[(size_t)ni * o + oi] * W[(size_t)oi * k + ki];
        }
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && ki < k) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][tx];
        const size_t idx = (size_t)ni * k + ki;
        if (silu_in) t *= silu_grad(in[idx]);
        din[idx] = accumulate ? din[idx] + t : t;
    }
}
hipError_t launch_linear_bwd_in_multi(const LinBwdJobs& J, const float* in, int n, int k, int o, int silu_in, float* din, int accumulate, hipStream_t s) {
    if (J.n < 1 || J.n > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(linear_bwd_in_multi_kernel, dim3((k + 63) / 64, n), dim3(1024), 0, s, J, in, n, k, o, silu_in, din, accumulate);
    return hipGetLastError();
}

hipError_t launch_linear_bwd_w(const float* in, const float* dout, int n, int k, int o, int silu_in, float* dW, float* db,
                               int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)((k + 63) / 64), (unsigned)((o + 7) / 8)), dim3(256), 0, s, in, dout, n, k, o, silu_in, dW, db, accumulate);
    return hipGetLastError();
}

// din[n][k] (+)= act'(in[n][k]) * sum_o dout[n][o] * W[o][k].  block = one item x 64 inputs x 16 output groups; fixed combination
// order: deterministic.  (8 items per block to share the W loads was measured slower: the loop is latency-, not traffic-bound.)
__global__ __launch_bounds__(1024) void linear_bwd_in_kernel(const float* in, const float* dout, const float* W, int n, int k, int o,
                                                             int silu_in, float* din, int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ni = blockIdx.y, ki = blockIdx.x * 64 + tx;
    float acc = 0.f;
    if (ki < k) {
#pragma unroll 8
        for (int oi = ty; oi < o; oi += 16) acc += dout[(size_t)ni * o + oi] * W[(size_t)oi * k + ki];
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && ki < k) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][tx];
        const size_t idx = (size_t)ni * k + ki;
        if (silu_in) t *= silu_grad(in[idx]);
        din[idx] = accumulate ? din[idx] + t : t;
    }
}

hipError_t launch_linear_bwd_in(const float* in, const float* dout, const float* W, int n, int k, int o, int silu_in,
                                float* din, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(linear_bwd_in_kernel, dim3((k + 63) / 64, n), dim3(1024), 0, s, in, dout, W, n, k, o, silu_in, din, accumulate);
    return hipGetLastError();
}

}  // namespace st
