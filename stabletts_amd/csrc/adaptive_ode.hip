// Device side of the adaptive Dormand-Prince 5(4) solver (torchdiffeq's default method, the reference's
// default at models/flow_matching.py:54 with solver=None): error / initial-step norms and the dense output.
// All fp32; reductions are two-stage with a fixed summation order, so step acceptance is reproducible.
#include "common.h"
#include "launch.h"

namespace st {

__global__ void set_scalar_kernel(float* dst, float v) { dst[0] = v; }

hipError_t launch_set_scalar(float* dst, float v, hipStream_t s) {
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, dst, v);
    return hipGetLastError();
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
    return r;
}

__global__ __launch_bounds__(256) void ode_norm_partial_kernel(const OdeNormArgs a) {
    __shared__ float sh[4];
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (int64_t)gridDim.x * 1024) {
        const float4 y = *(const float4*)(a.y + i);
        const float yy[4] = {y.x, y.y, y.z, y.w};
        if (a.mode == 0) {
            const float4 f = *(const float4*)(a.b + i);
            const float ff[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sc = a.atol + a.rtol * fabsf(yy[e]);
                const float q0 = yy[e] / sc, q1 = ff[e] / sc;
                s0 += q0 * q0; s1 += q1 * q1;
            }
        } else if (a.mode == 1) {
            const float4 f0 = *(const float4*)(a.a + i);
            const float4 f1 = *(const float4*)(a.b + i);
            const float dd[4] = {f1.x - f0.x, f1.y - f0.y, f1.z - f0.z, f1.w - f0.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = dd[e] / (a.atol + a.rtol * fabsf(yy[e]));
                s0 += q * q;
            }
        } else if (a.mode == 3) {
            // implicit Adams corrector: number of elements whose |a - b| / (atol + rtol max(|a|, |b|)) is not < 1
            // (torchdiffeq's max-norm convergence test: converged iff the count is 0; NaN counts)
            const float4 p0 = *(const float4*)(a.a + i), p1 = *(const float4*)(a.b + i);
            const float av[4] = {p0.x, p0.y, p0.z, p0.w}, bv[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = fabsf(av[e] - bv[e]) / (a.atol + a.rtol * fmaxf(fabsf(av[e]), fabsf(bv[e])));
                s0 += (q < 1.0f) ? 0.f : 1.f;
            }
        } else {
            const float4 y1 = *(const float4*)(a.a + i);
            const float y1v[4] = {y1.x, y1.y, y1.z, y1.w};
            float er[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if (j < a.nk) {
                    const float4 kv = *(const float4*)(a.k[j] + i);
                    const float c = a.coef[j];
                    er[0] += c * kv.x; er[1] += c * kv.y; er[2] += c * kv.z; er[3] += c * kv.w;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = er[e] / (a.atol + a.rtol * fmaxf(fabsf(yy[e]), fabsf(y1v[e])));
                s0 += q * q;
            }
        }
    }
    const float t0 = block_sum(s0, sh);
    const float t1 = block_sum(s1, sh);
    if (threadIdx.x == 0) { a.partial[2 * blockIdx.x] = t0; a.partial[2 * blockIdx.x + 1] = t1; }
}

__global__ __launch_bounds__(256) void ode_norm_final_kernel(const float* partial, int nblocks, float* out) {
    __shared__ float sh[4];
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) { s0 += partial[2 * i]; s1 += partial[2 * i + 1]; }
    const float t0 = block_sum(s0, sh);
    const float t1 = block_sum(s1, sh);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; }
}

hipError_t launch_ode_norm(const OdeNormArgs& a, hipStream_t s) {
    int64_t want = (a.n / 4 + 255) / 256;
    const int grid = (int)(want < 1 ? 1 : (want > kOdeNormBlocks ? kOdeNormBlocks : want));
    hipLaunchKernelGGL(ode_norm_partial_kernel, dim3(grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(ode_norm_final_kernel, dim3(1), dim3(256), 0, s, a.partial, grid, a.out);
    return hipGetLastError();
}

struct InterpArgs { const float* y0; const float* y1; const float* k[7]; float cmid_dt[7]; float dt, x; int64_t n; float* out; };

__global__ __launch_bounds__(256) void dopri5_interp_kernel(const InterpArgs a) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= a.n) return;
    const float4 y0 = *(const float4*)(a.y0 + i), y1 = *(const float4*)(a.y1 + i);
    const float4 f0 = *(const float4*)(a.k[0] + i), f1 = *(const float4*)(a.k[6] + i);
    float ym[4] = {y0.x, y0.y, y0.z, y0.w};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const float4 kv = *(const float4*)(a.k[j] + i);
        const float c = a.cmid_dt[j];
        ym[0] += c * kv.x; ym[1] += c * kv.y; ym[2] += c * kv.z; ym[3] += c * kv.w;
    }
    const float Y0[4] = {y0.x, y0.y, y0.z, y0.w}, Y1[4] = {y1.x, y1.y, y1.z, y1.w};
    const float F0[4] = {f0.x, f0.y, f0.z, f0.w}, F1[4] = {f1.x, f1.y, f1.z, f1.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ca = 2.f * a.dt * (F1[e] - F0[e]) - 8.f * (Y1[e] + Y0[e]) + 16.f * ym[e];
        const float cb = a.dt * (5.f * F0[e] - 3.f * F1[e]) + 18.f * Y0[e] + 14.f * Y1[e] - 32.f * ym[e];
        const float cc = a.dt * (F1[e] - 4.f * F0[e]) - 11.f * Y0[e] - 5.f * Y1[e] + 16.f * ym[e];
        const float cd = a.dt * F0[e];
        o[e] = (((ca * a.x + cb) * a.x + cc) * a.x + cd) * a.x + Y0[e];
    }
    *(float4*)(a.out + i) = make_float4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_dopri5_interp(const float* y0, const float* y1, const float* const* k, const float* cmid_dt, float dt,
                                float x, int64_t n, float* out, hipStream_t s) {
    InterpArgs a;
    a.y0 = y0; a.y1 = y1; a.dt = dt; a.x = x; a.n = n; a.out = out;
    for (int j = 0; j < 7; ++j) { a.k[j] = k[j]; a.cmid_dt[j] = cmid_dt[j]; }
    const int grid = (int)((n / 4 + 255) / 256);
    hipLaunchKernelGGL(dopri5_interp_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace st
