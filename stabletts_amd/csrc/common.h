// Device-side common types for the gfx950 CFM-decoder kernels.
// MFMA operand types (bf16 / f16) are a template policy; accumulation is always fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace st {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// 32x32x16 MFMA (gfx950): D[i][j] += sum_k A[i][k] B[k][j]
//   A operand: lane l holds A[i = l&31][k-slots (l>>5)*8 .. +8]
//   B operand: lane l holds B[k-slots (l>>5)*8 .. +8][j = l&31]
//   C/D      : lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
struct OpBF16 {
    using elem = __bf16;
    using vec8 = bf16x8_t;
    static __device__ __forceinline__ f32x16_t mfma(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
struct OpF16 {
    using elem = _Float16;
    using vec8 = f16x8_t;
    static __device__ __forceinline__ f32x16_t mfma(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <class P>
__device__ __forceinline__ typename P::elem to16(float f) { return (typename P::elem)f; }

// pack 4 floats into 4 16-bit values (8 bytes), round-to-nearest-even
template <class P>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) typename P::elem v4;
    v4 v;
    v[0] = (typename P::elem)a; v[1] = (typename P::elem)b;
    v[2] = (typename P::elem)c; v[3] = (typename P::elem)d;
    return __builtin_bit_cast(uint2, v);
}

template <class P>
__device__ __forceinline__ typename P::vec8 as_vec8(uint4 v) {
    return __builtin_bit_cast(typename P::vec8, v);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// SiLU on the hardware transcendental units (v_exp_f32 + v_rcp_f32, ~1 ulp each): used where the
// result is rounded to a 16-bit MFMA operand anyway.  exp2 overflow (x << 0) gives rcp(inf) = 0.
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int kLdsRowBytes = 144;  // 64 x 16-bit channels + 16 B pad: conflict-free ds_read_b128 over 16 rows

}  // namespace st
