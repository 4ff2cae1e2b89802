// Duration -> alignment -> mu_y (length regulation) of StableTTS.synthesise, models/model.py:82-96 + generate_path
// (:17-27): integer / index work, HBM-bound and tiny.  Everything follows the reference's fp32 arithmetic so that
// w_ceil, y_lengths, the 0/1 alignment and the gathered mu_y are bit-exact for exactly summable durations
// (length_scale = 1 or any dyadic value; the cumulative sum is sequential fp32 like torch.cumsum on the CPU).
#include "common.h"
#include "launch.h"

namespace st {

// one block per utterance: w_ceil[i] = ceil(exp(logw[i]) * mask[i]) * length_scale ; cum = cumsum(w_ceil) ;
// y_len = (int64) max(sum, 1)
__global__ __launch_bounds__(256) void durations_kernel(const float* logw, const float* x_mask, float length_scale, int Tx,
                                                        float* w_ceil, float* cum, long long* y_len) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < Tx; i += blockDim.x) {
        const float w = expf(logw[(size_t)b * Tx + i]) * x_mask[(size_t)b * Tx + i];
        w_ceil[(size_t)b * Tx + i] = ceilf(w) * length_scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int i = 0; i < Tx; ++i) { acc += w_ceil[(size_t)b * Tx + i]; cum[(size_t)b * Tx + i] = acc; }
        y_len[b] = (long long)fmaxf(acc, 1.0f);
    }
}

hipError_t launch_durations(const float* logw, const float* x_mask, float length_scale, int B, int Tx, float* w_ceil,
                            float* cum, long long* y_len, hipStream_t s) {
    hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(256), 0, s, logw, x_mask, length_scale, Tx, w_ceil, cum, y_len);
    return hipGetLastError();
}

// cum from a given duration tensor (generate_path's own first step, models/model.py:19)
__global__ void cumsum_rows_kernel(const float* dur, int Tx, float* cum) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    float acc = 0.f;
    for (int i = 0; i < Tx; ++i) { acc += dur[(size_t)b * Tx + i]; cum[(size_t)b * Tx + i] = acc; }
}
hipError_t launch_cumsum_rows(const float* dur, int B, int Tx, float* cum, hipStream_t s) {
    hipLaunchKernelGGL(cumsum_rows_kernel, dim3(B), dim3(64), 0, s, dur, Tx, cum);
    return hipGetLastError();
}

// path[b][i][j] = ((j < cum[i]) - (j < cum[i-1])) * mask[b][i][j]   (models/model.py:22-26; mask may be any (B,Tx,Ty) tensor)
__global__ __launch_bounds__(256) void path_kernel(const float* cum, const float* mask, int Tx, int Ty, float* path) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Ty) return;
    const float hi = cum[(size_t)b * Tx + i];
    const float lo = i > 0 ? cum[(size_t)b * Tx + i - 1] : 0.f;
    const float fj = (float)j;
    const float v = (fj < hi ? 1.f : 0.f) - (fj < lo ? 1.f : 0.f);
    const size_t o = ((size_t)b * Tx + i) * Ty + j;
    path[o] = v * mask[o];
}
hipError_t launch_path(const float* cum, const float* mask, int B, int Tx, int Ty, float* path, hipStream_t s) {
    hipLaunchKernelGGL(path_kernel, dim3((Ty + 255) / 256, Tx, B), dim3(256), 0, s, cum, mask, Tx, Ty, path);
    return hipGetLastError();
}

// Length regulation: frame j of utterance b copies text position i(j) = the first i with j < cum[i] (binary search),
// provided j < y_len[b] and x_mask[i] != 0 -- exactly the single non-zero column of the reference's 0/1 alignment,
// so mu_y = attn^T mu_x (model.py:94) becomes a gather.  Also writes y_mask and (optionally) the alignment itself.
__global__ __launch_bounds__(256) void align_kernel(const float* cum, const float* x_mask, const long long* y_len,
                                                    const float* mu_x, int M, int Tx, int Ty, float* attn, float* mu_y,
                                                    float* y_mask) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Ty) return;
    const float* c = cum + (size_t)b * Tx;
    const float fj = (float)j;
    int lo = 0, hi = Tx;                    // first i with fj < c[i]
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (fj < c[mid]) hi = mid; else lo = mid + 1; }
    const bool inside = (long long)j < y_len[b];
    const bool hit = inside && lo < Tx && x_mask[(size_t)b * Tx + lo] != 0.f;
    if (y_mask) y_mask[(size_t)b * Ty + j] = inside ? 1.f : 0.f;
    const float xm = hit ? x_mask[(size_t)b * Tx + lo] : 0.f;
    for (int m = 0; m < M; ++m)
        mu_y[((size_t)b * M + m) * Ty + j] = hit ? mu_x[((size_t)b * M + m) * Tx + lo] * xm : 0.f;
    if (attn)
        for (int i = 0; i < Tx; ++i) attn[((size_t)b * Tx + i) * Ty + j] = (hit && i == lo) ? xm : 0.f;
}
hipError_t launch_align(const float* cum, const float* x_mask, const long long* y_len, const float* mu_x, int B, int M, int Tx,
                        int Ty, float* attn, float* mu_y, float* y_mask, hipStream_t s) {
    hipLaunchKernelGGL(align_kernel, dim3((Ty + 255) / 256, B), dim3(256), 0, s, cum, x_mask, y_len, mu_x, M, Tx, Ty, attn, mu_y, y_mask);
    return hipGetLastError();
}

}  // namespace st
