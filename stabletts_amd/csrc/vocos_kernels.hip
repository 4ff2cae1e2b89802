// Vocos vocoder: the kernels around its GEMMs (vocoders/vocos/models/*.py; config.py:46-50 -> 128 -> 512, 8 ConvNeXt
// blocks, ISTFT head n_fft 2048 / hop 512).  Everything here is row-wise HBM-bound work on time-major tensors
// [utterance][frame][channel]; the five GEMM shapes run on the implicit-GEMM kernels of conv_gemm2_impl.h.
#include "common.h"
#include "vocos_launch.h"

namespace st {

// ---------------------------------------------------------------- embed: im2col of the k = 7 convolution (backbone.py:28,51)
// One block = 64 frames of one utterance: the (M x 70) mel tile is read with frames contiguous and written as
// 16-bit rows [frame][tap][channel].
template <class P>
__global__ __launch_bounds__(256) void voc_im2col7_kernel(const float* __restrict__ mel, int M, int T, typename P::elem* __restrict__ a16) {
    extern __shared__ float tile[];           // [M][72]
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const float* src = mel + (size_t)b * M * T;
    for (int i = threadIdx.x; i < M * 70; i += 256) {
        const int c = i / 70, k = i - c * 70;
        const int t = t0 + k - 3;
        tile[c * 72 + k] = (t >= 0 && t < T) ? src[(size_t)c * T + t] : 0.0f;
    }
    __syncthreads();
    const int K = 7 * M;
    for (int i = threadIdx.x; i < 64 * K; i += 256) {
        const int f = i / K, r = i - f * K;
        const int j = r / M, c = r - j * M;
        if (t0 + f < T) a16[((size_t)b * T + t0 + f) * K + r] = to16<P>(tile[c * 72 + f + j]);
    }
}

hipError_t launch_voc_im2col7(int dtype, const float* mel, int B, int M, int T, void* a16, hipStream_t s) {
    const dim3 grid((T + 63) / 64, B), blk(256);
    const size_t lds = (size_t)M * 72 * 4;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (dtype == DT_BF16) hipLaunchKernelGGL((voc_im2col7_kernel<OpBF16>), grid, blk, lds, s, mel, M, T, (OpBF16::elem*)a16);
    else                  hipLaunchKernelGGL((voc_im2col7_kernel<OpF16>), grid, blk, lds, s, mel, M, T, (OpF16::elem*)a16);
    return hipGetLastError();
}

// ---------------------------------------------------------------- LayerNorm(512, eps 1e-6, affine): one wave per row, lane = 8 channels
struct Row8 { float4 a, b; };
__device__ __forceinline__ Row8 ld8(const float* p) { Row8 r; r.a = *(const float4*)p; r.b = *(const float4*)(p + 4); return r; }
__device__ __forceinline__ void voc_ln8(Row8& v, const Row8& w, const Row8& b) {
    const float mean = wave_sum(v.a.x + v.a.y + v.a.z + v.a.w + v.b.x + v.b.y + v.b.z + v.b.w) * (1.0f / 512.0f);
    v.a.x -= mean; v.a.y -= mean; v.a.z -= mean; v.a.w -= mean; v.b.x -= mean; v.b.y -= mean; v.b.z -= mean; v.b.w -= mean;
    const float var = wave_sum(v.a.x * v.a.x + v.a.y * v.a.y + v.a.z * v.a.z + v.a.w * v.a.w +
                               v.b.x * v.b.x + v.b.y * v.b.y + v.b.z * v.b.z + v.b.w * v.b.w) * (1.0f / 512.0f);
    const float rs = 1.0f / sqrtf(var + 1e-6f);
    v.a.x = v.a.x * rs * w.a.x + b.a.x; v.a.y = v.a.y * rs * w.a.y + b.a.y; v.a.z = v.a.z * rs * w.a.z + b.a.z; v.a.w = v.a.w * rs * w.a.w + b.a.w;
    v.b.x = v.b.x * rs * w.b.x + b.b.x; v.b.y = v.b.y * rs * w.b.y + b.b.y; v.b.z = v.b.z * rs * w.b.z + b.b.z; v.b.w = v.b.w * rs * w.b.w + b.b.w;
}
template <class P>
__device__ __forceinline__ void st16x8(void* p, const Row8& v) {
    const uint2 lo = pack4<P>(v.a.x, v.a.y, v.a.z, v.a.w), hi = pack4<P>(v.b.x, v.b.y, v.b.z, v.b.w);
    *(uint4*)p = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

template <class P>
__global__ __launch_bounds__(256) void voc_ln_kernel(const float* x, const float* __restrict__ w, const float* __restrict__ b,
                                                     long long rows, float* out32, void* out16, void* out16_lo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int ch = lane * 8;
    Row8 v = ld8(x + (size_t)row * 512 + ch);
    voc_ln8(v, ld8(w + ch), ld8(b + ch));
    if (out32) { *(float4*)(out32 + (size_t)row * 512 + ch) = v.a; *(float4*)(out32 + (size_t)row * 512 + ch + 4) = v.b; }
    if (out16) st16x8<P>((unsigned char*)out16 + ((size_t)row * 512 + ch) * 2, v);
    if (out16_lo) {      // the low half of a split-precision operand pair: x - float(round16(x)), rounded once more
        Row8 r;
        r.a.x = v.a.x - (float)to16<P>(v.a.x); r.a.y = v.a.y - (float)to16<P>(v.a.y); r.a.z = v.a.z - (float)to16<P>(v.a.z); r.a.w = v.a.w - (float)to16<P>(v.a.w);
        r.b.x = v.b.x - (float)to16<P>(v.b.x); r.b.y = v.b.y - (float)to16<P>(v.b.y); r.b.z = v.b.z - (float)to16<P>(v.b.z); r.b.w = v.b.w - (float)to16<P>(v.b.w);
        st16x8<P>((unsigned char*)out16_lo + ((size_t)row * 512 + ch) * 2, r);
    }
}

hipError_t launch_voc_ln(int dtype, const float* x, const float* w, const float* b, int64_t rows, float* out32, void* out16,
                         void* out16_lo, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    if (dtype == DT_BF16) hipLaunchKernelGGL((voc_ln_kernel<OpBF16>), grid, blk, 0, s, x, w, b, (long long)rows, out32, out16, out16_lo);
    else                  hipLaunchKernelGGL((voc_ln_kernel<OpF16>), grid, blk, 0, s, x, w, b, (long long)rows, out32, out16, out16_lo);
    return hipGetLastError();
}

// ---------------------------------------------------------------- ConvNeXt block prologue: depthwise k = 7 conv + LayerNorm (module.py:35-37)
// One wave per R consecutive frames of one utterance (R = 4; 1 for small batches), lane = 8 channels: the R + 6 fp32 residual
// rows of the window are requested up front and the 56 taps of the lane's channels loaded once per wave.
template <class P, int R>
__global__ __launch_bounds__(256) void voc_dwconv_ln_kernel(const float* __restrict__ x, const float* __restrict__ dw,
                                                            const float* __restrict__ dbias, const float* __restrict__ w,
                                                            const float* __restrict__ b, int T, int groups_per_item, long long n_groups,
                                                            void* h16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long grp = (long long)blockIdx.x * 4 + wave;
    if (grp >= n_groups) return;
    const int item = (int)(grp / groups_per_item);
    const int t0 = (int)(grp - (long long)item * groups_per_item) * R;
    const int ch = lane * 8;
    const float* xi = x + (size_t)item * T * 512 + ch;
    const Row8 zero = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    Row8 win[R + 6];          // every row of the window is requested before anything is computed: one exposed latency per wave
#pragma unroll
    for (int j = 0; j < R + 6; ++j) {
        const int t = t0 - 3 + j;
        win[j] = (t >= 0 && t < T) ? ld8(xi + (size_t)t * 512) : zero;      // zero padding of nn.Conv1d(padding=3)
    }
    float wt[8][7];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) wt[i][j] = dw[(size_t)(ch + i) * 7 + j];
    const Row8 bias = ld8(dbias + ch), lw = ld8(w + ch), lb = ld8(b + ch);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = t0 + r;
        if (t >= T) break;                          // wave-uniform
        Row8 acc = bias;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const Row8& v = win[r + j];
            acc.a.x += wt[0][j] * v.a.x; acc.a.y += wt[1][j] * v.a.y; acc.a.z += wt[2][j] * v.a.z; acc.a.w += wt[3][j] * v.a.w;
            acc.b.x += wt[4][j] * v.b.x; acc.b.y += wt[5][j] * v.b.y; acc.b.z += wt[6][j] * v.b.z; acc.b.w += wt[7][j] * v.b.w;
        }
        voc_ln8(acc, lw, lb);
        st16x8<P>((unsigned char*)h16 + (((size_t)item * T + t) * 512 + ch) * 2, acc);
    }
}

template <class P, int R>
static void launch_dw(const float* x, const float* dw, const float* dbias, const float* w, const float* b, int B, int T, void* h16, hipStream_t s) {
    const int gpi = (T + R - 1) / R;
    const long long n_groups = (long long)B * gpi;
    hipLaunchKernelGGL((voc_dwconv_ln_kernel<P, R>), dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0, s, x, dw, dbias, w, b, T, gpi, n_groups, h16);
}

hipError_t launch_voc_dwconv_ln(int dtype, const float* x, const float* dw, const float* dbias, const float* w, const float* b,
                                int B, int T, void* h16, hipStream_t s) {
    const bool big = (long long)B * T >= 8192;      // small batches: one frame per wave (more waves than a few CUs' worth)
    if (dtype == DT_BF16) { if (big) launch_dw<OpBF16, 4>(x, dw, dbias, w, b, B, T, h16, s); else launch_dw<OpBF16, 1>(x, dw, dbias, w, b, B, T, h16, s); }
    else                  { if (big) launch_dw<OpF16, 4>(x, dw, dbias, w, b, B, T, h16, s); else launch_dw<OpF16, 1>(x, dw, dbias, w, b, B, T, h16, s); }
    return hipGetLastError();
}

// ---------------------------------------------------------------- ISTFT head (head.py:103-116, ISTFT.forward :50-57)
// One block (256 threads) per frame.  S[k] = min(exp(m_k), 100) (cos p_k + i sin p_k), k = 0..1024, then the real
// inverse FFT of length 2048 as ONE complex inverse FFT of length 1024:
//     E[k] = (S[k] + conj(S[1024-k])) / 2,   O[k] = (S[k] - conj(S[1024-k])) / 2 * e^{+2 pi i k / 2048},   Z = E + i O,
//     z = IDFT_1024(Z):   x[2n] = Re z[n],  x[2n+1] = Im z[n]
// (imaginary parts of the DC and Nyquist bins ignored, as a complex-to-real transform does).  The 1024-point
// transform is 5 radix-4 Stockham autosort passes between two LDS buffers; twiddles from an LDS table built per block.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(256) void voc_spec_ifft_kernel(const float* __restrict__ head, const float* __restrict__ window,
                                                            float* __restrict__ frames) {
    __shared__ float2 bufA[1024 + 8], bufB[1024 + 8], tw[1024];
    const int tid = threadIdx.x;
    const float* row = head + (size_t)blockIdx.x * (2 * kVocHeadPlane);
    for (int k = tid; k <= 1024; k += 256) {
        const float mag = fminf(expf(row[k]), 100.0f);           // head.py:105-106
        float sn, cs;
        sincosf(row[kVocHeadPlane + k], &sn, &cs);               // :108-109
        bufB[k] = make_float2(mag * cs, (k == 0 || k == 1024) ? 0.0f : mag * sn);
    }
    for (int k = tid; k < 1024; k += 256) {
        float sn, cs;
        sincospif((float)k * (1.0f / 512.0f), &sn, &cs);         // e^{+2 pi i k / 1024}
        tw[k] = make_float2(cs, sn);
    }
    __syncthreads();
    for (int k = tid; k < 1024; k += 256) {
        const float2 a = bufB[k], c = bufB[1024 - k];
        const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
        float2 O = make_float2(0.5f * (a.x - c.x), 0.5f * (a.y + c.y));
        float sn, cs;
        sincospif((float)k * (1.0f / 1024.0f), &sn, &cs);        // e^{+2 pi i k / 2048}
        O = cmul(O, make_float2(cs, sn));
        bufA[k] = make_float2(E.x - O.y, E.y + O.x);             // E + i O
    }
    __syncthreads();
    float2* in = bufA; float2* out = bufB;
#pragma unroll 1
    for (int Ns = 1; Ns < 1024; Ns <<= 2) {
        const int j = tid, k = j & (Ns - 1);
        const int tstep = k * (256 / Ns);                         // twiddle exponent of t = 1: k / (4 Ns) turns = k * 256 / Ns / 1024
        const float2 u0 = in[j];
        const float2 u1 = cmul(in[j + 256], tw[tstep]);
        const float2 u2 = cmul(in[j + 512], tw[2 * tstep]);
        const float2 u3 = cmul(in[j + 768], tw[3 * tstep]);
        const float2 s02 = make_float2(u0.x + u2.x, u0.y + u2.y), d02 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 s13 = make_float2(u1.x + u3.x, u1.y + u3.y), d13 = make_float2(u1.x - u3.x, u1.y - u3.y);
        const int j0 = ((j - k) << 2) + k;
        out[j0] = make_float2(s02.x + s13.x, s02.y + s13.y);
        out[j0 + Ns] = make_float2(d02.x - d13.y, d02.y + d13.x);       // u0 + i u1 - u2 - i u3
        out[j0 + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
        out[j0 + 3 * Ns] = make_float2(d02.x + d13.y, d02.y - d13.x);   // u0 - i u1 - u2 + i u3
        __syncthreads();
        float2* tmp = in; in = out; out = tmp;
    }
    float* dst = frames + (size_t)blockIdx.x * kVocNfft;
    for (int n = tid; n < 1024; n += 256) {
        const float2 z = in[n];
        const float2 wv = *(const float2*)(window + 2 * n);
        *(float2*)(dst + 2 * n) = make_float2(z.x * (1.0f / 1024.0f) * wv.x, z.y * (1.0f / 1024.0f) * wv.y);   // :56-57
    }
}

hipError_t launch_voc_spec_ifft(const float* head, const float* window, int64_t rows, float* frames, hipStream_t s) {
    hipLaunchKernelGGL(voc_spec_ifft_kernel, dim3((unsigned)rows), dim3(256), 0, s, head, window, frames);
    return hipGetLastError();
}

// Overlap-add (fold, head.py:60-63), "same" trim (:48,63), window envelope (:66-69) and normalisation (:73).
// Output sample s of utterance b sits at q = s + 768 of the untrimmed signal; frames floor((q - 2047 + 511) / 512) ..
// floor(q / 512) cover it (at most 4), added in ascending frame order.
__global__ __launch_bounds__(256) void voc_overlap_add_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                              int T, float* __restrict__ audio) {
    const long long len = (long long)T * kVocHop;
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s >= len) return;
    const int b = blockIdx.y;
    const long long q = s + (kVocNfft - kVocHop) / 2;
    int f1 = (int)(q / kVocHop); if (f1 > T - 1) f1 = T - 1;
    long long f0l = (q - (kVocNfft - 1) + kVocHop - 1) / kVocHop; if (q - (kVocNfft - 1) < 0) f0l = 0;
    float y = 0.0f, env = 0.0f;
    for (int f = (int)f0l; f <= f1; ++f) {
        const int off = (int)(q - (long long)f * kVocHop);
        const float wv = window[off];
        y += frames[((size_t)b * T + f) * kVocNfft + off];
        env += wv * wv;
    }
    audio[(size_t)b * len + s] = y / env;
}

hipError_t launch_voc_overlap_add(const float* frames, const float* window, int B, int T, float* audio, hipStream_t s) {
    const long long len = (long long)T * kVocHop;
    hipLaunchKernelGGL(voc_overlap_add_kernel, dim3((unsigned)((len + 255) / 256), B), dim3(256), 0, s, frames, window, T, audio);
    return hipGetLastError();
}

}  // namespace st
