// Device-side common types for the gfx950 CFM-decoder kernels.
// MFMA operand types (bf16 / f16) are a template policy; accumulation is always fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace st {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// 32x32x16 MFMA (gfx950): D[i][j] += sum_k A[i][k] B[k][j]
//   A operand: lane l holds A[i = l&31][k-slots (l>>5)*8 .. +8]
//   B operand: lane l holds B[k-slots (l>>5)*8 .. +8][j = l&31]
//   C/D      : lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
// 16x16x32 MFMA (mfma16; the same FLOPs per cycle as 32x32x16 on zeros, ~17 % more sustained throughput on real data: the chip is
// power-limited in the MFMA loops and this shape moves 20 % fewer register-file bytes per FLOP, profiles/r04_mfma_power_shapes.txt):
//   A operand: lane l holds A[i = l&15][k-slots (l>>4)*8 .. +8]      B operand: lane l holds B[k-slots (l>>4)*8 .. +8][j = l&15]
//   C/D      : lane l, reg r holds D[i = 4*(l>>4) + r][j = l&15]
struct OpBF16 {
    using elem = __bf16;
    using vec8 = bf16x8_t;
    static constexpr float kMaskedScore = -1e30f;     // 16-bit representable "minus infinity" of a masked attention score
    static __device__ __forceinline__ f32x16_t mfma(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
struct OpF16 {
    using elem = _Float16;
    using vec8 = f16x8_t;
    static constexpr float kMaskedScore = -60000.0f;  // f16 saturates at 65504; exp2(-60000 + anything sane) == 0
    static __device__ __forceinline__ f32x16_t mfma(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
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

// pack4 of four values times a common scale, every product rounded to fp32 FIRST: with -ffp-contract=fast hipcc fuses
// (f16)(x * s) into v_fma_mixlo_f16 (one rounding) at some call sites and not at others -- the q/k/v kernels must agree bit for bit
template <class P>
__device__ __forceinline__ uint2 scale_pack4(float a, float b, float c, float d, float s) {
    float pa = a * s, pb = b * s, pc = c * s, pd = d * s;
    asm volatile("" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));      // (the fold is an instruction-selection pattern: pragmas do not stop it)
    return pack4<P>(pa, pb, pc, pd);
}

// ... and the rounding residuals of the same four products: lo = fl16(x s - fl16(x s)) (split-precision q / k operands: hi + lo
// carries ~22 bits of the fp32 value; the hi part is bit-identical to scale_pack4's)
template <class P>
__device__ __forceinline__ uint2 scale_pack4_lo(float a, float b, float c, float d, float s) {
    float pa = a * s, pb = b * s, pc = c * s, pd = d * s;
    asm volatile("" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));
    return pack4<P>(pa - (float)to16<P>(pa), pb - (float)to16<P>(pb), pc - (float)to16<P>(pc), pd - (float)to16<P>(pd));
}

template <class P>
__device__ __forceinline__ typename P::vec8 as_vec8(uint4 v) {
    return __builtin_bit_cast(typename P::vec8, v);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_grad(float a) {      // d/da [a * sigmoid(a)] = s * (1 + a * (1 - s))
    const float s = 1.0f / (1.0f + expf(-a));
    return s * (1.0f + a * (1.0f - s));
}

// SiLU on the hardware transcendental units (v_exp_f32 + v_rcp_f32, ~1 ulp each): used where the
// result is rounded to a 16-bit MFMA operand anyway.  exp2 overflow (x << 0) gives rcp(inf) = 0.
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float silu_grad_fast(float a) {      // silu_grad on the same two hardware units
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a));
    return s * (1.0f + a * (1.0f - s));
}

// Cross-lane reductions on the VALU (DPP row rotations + gfx950 v_permlane{16,32}_swap): __shfl_xor lowers to
// ds_bpermute_b32, an LDS-crossbar round trip of ~60+ cycles per step on the dependent chain.
// nn.GELU() (exact erf form) with erf from Abramowitz & Stegun 7.1.26: branch-free, |error| <= 1.5e-7 on erf, 4.7e-7 absolute on
// GELU over [-8, 8] (fp32 evaluation) -- far below the 16-bit rounding of the activation it produces; libm's erff is a
// piecewise evaluation whose branches both run under a divergent wave.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float er = copysignf(fmaf(-poly, e, 1.0f), z);
    return 0.5f * x * (1.0f + er);
}

// fp32 add of the attention kernels' row sums.  hipcc's SLP vectoriser would fuse neighbouring adds into v_pk_add_f32, and beside
// MFMAs a packed fp32 op costs more than the two scalar ones it replaces (MI355X_MICROARCH.md, per-instruction constants;
// measured here: attention +28 %) -- so attention.hip is compiled with -fno-slp-vectorize (build.py) and this is a plain add.
// Until round 4 it was an inline-asm v_add_f32: hipcc does not run its hazard recogniser over asm operands, and where it
// scheduled the add directly behind the v_exp_f32 that produces its input (the rarely taken re-reference path of the inference
// kernel) the add read the register before the transcendental unit had written it (gfx950 needs one wait state there): row sums
// off by the stale value, outputs scaled by up to 4x for moderately peaky attention (profiles/r04_attention_trans_hazard.txt).
__device__ __forceinline__ float add_f32_scalar(float a, float b) { return a + b; }
// The training variant of the attention kernel sits exactly at 128 VGPRs and spills with the plain add: it keeps the asm form,
// with the wait state the hazard needs INSIDE the statement (s_nop 0 = one wait state: the transcendental result is readable).
__device__ __forceinline__ float add_f32_asm_safe(float a, float b) {
    float r;
    asm("s_nop 0\n\tv_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Partial-RoPE rotation of one pair (x1, x2) by (c, s) with a FIXED contraction (one fma per output: x1 c - x2 s = fma(x1, c, -(x2 s)),
// x2 c + x1 s = fma(x2, c, x1 s)): the two q/k/v kernels (conv_gemm2_impl.h g2_epilogue_qkv, qkv_ws.hip) must round identically,
// and left to -ffp-contract=fast hipcc picks the fused product per call site.
__device__ __forceinline__ void rope_rot(float& x1, float& x2, float c, float s) {
#pragma clang fp contract(off)
    const float p = x2 * s, q = x1 * s;
    const float a = __builtin_fmaf(x1, c, -p), b = __builtin_fmaf(x2, c, q);
    x1 = a; x2 = b;
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
// value of lane^32 combined with the lane's own: both halves end up with op(v[l], v[l^32])
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// sum over the 64 lanes, result in every lane (deterministic order)
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x128>(v); v = dpp_add<0x124>(v); v = dpp_add<0x122>(v); v = dpp_add<0x121>(v);   // row_ror 8,4,2,1
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return xor32_sum(__uint_as_float(a[0]) + __uint_as_float(a[1]));
}

// Write-through (agent-scope, sc1) global stores for the coalesced epilogues.  MI355X has one L2 per XCD and the
// consumer of an activation tensor is always the NEXT kernel, running on all XCDs: with ordinary stores up to
// 8 x 4 MB of dirty lines sit in the L2s until the end-of-kernel release writes them back -- measured as a
// ~10 us idle gap after every large kernel (rocprofv3 kernel trace) -- whereas written through they reach HBM
// while the kernel is still computing.  Only for full-line coalesced rows (partial-line scatter stores, e.g.
// the attention output, keep the write-back path so the L2 can merge them).
// The trailing s_nop covers the store-data hazard that hipcc cannot see behind inline asm (guide 5.7).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#if !defined(ST_ROW_STORE_POLICY) && defined(ST_ROW_STORE_POL)      // A/B builds: -DST_ROW_STORE_POL=n
#if ST_ROW_STORE_POL == 1
#define ST_ROW_STORE_POLICY "nt"
#elif ST_ROW_STORE_POL == 2
#define ST_ROW_STORE_POLICY "sc1 nt"
#elif ST_ROW_STORE_POL == 3
#define ST_ROW_STORE_POLICY "sc0 sc1"
#elif ST_ROW_STORE_POL == 4
#define ST_ROW_STORE_POLICY "sc0 sc1 nt"
#endif
#endif
#ifndef ST_ROW_STORE_POLICY
#define ST_ROW_STORE_POLICY "sc1"
#endif
__device__ __forceinline__ void store_row16(void* p, uint4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off " ST_ROW_STORE_POLICY "\n\ts_nop 1" :: "v"(p), "v"(__builtin_bit_cast(u32x4_t, v)) : "memory");
}
__device__ __forceinline__ void store_row16(void* p, float4 v) { store_row16(p, __builtin_bit_cast(uint4, v)); }
__device__ __forceinline__ void store_row8(void* p, uint2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off " ST_ROW_STORE_POLICY "\n\ts_nop 1" :: "v"(p), "v"(__builtin_bit_cast(u32x2_t, v)) : "memory");
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

// LDS-DMA of one 1-KiB piece (64 lanes x 16 B), issued from INLINE ASM on purpose: with the
// __builtin_amdgcn_global_load_lds form hipcc (ROCm 7.2) treats every later ds_read as possibly aliasing the
// in-flight DMA and emits `s_waitcnt vmcnt(0)` in front of the first fragment read of the stage -- the DMA
// just issued is drained before any MFMA starts, i.e. copy and compute never overlap inside a wave (seen in
// the ISA; measured: DMA-only 113 us + compute-only 97 us -> 142 us).  An asm statement is invisible to that
// pass; completion is tracked by hand with counted `s_waitcnt vmcnt(N)` in front of the stage barrier.
// M0 (LDS destination base of the DMA) is saved/restored inside the statement (guide section 5.7).
__device__ __forceinline__ void glds16b(const void* gsrc, unsigned char* lds_wave_base) {
    const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(off) : "memory");
}
// same transfer with the address split as SGPR base + 32-bit per-lane offset (global saddr form): the per-lane
// part is loop invariant and precomputed once, the per-stage part is scalar arithmetic
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned char* lds_wave_base) {
    const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(off) : "memory");
}
// the same transfer with the LDS destination given as a wave-uniform LDS byte offset (no generic -> LDS pointer cast, no
// readfirstlane: scalar arithmetic only -- on a SIMD every VALU instruction of a K loop is issued instead of an MFMA)
__device__ __forceinline__ void glds16o(const void* sbase, unsigned voff, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void glds16bo(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}
// (Round 5 measured the same piece with the non-temporal hint, `global_load_lds_dwordx4 ... nt`, for the activation tiles a block streams
// once: -1.1 % on the single-sequence solve, 0.0 % on the default two-part solve, +1.5 % (slower) on a ragged batch, +8 % on the q/k/v
// projection's own tile -- profiles/r05_ab_nt_dma.txt, r05_ab_store_policy.txt.  Not kept.)
// the 4-byte-per-lane form (64 lanes x 4 B = 256 B at lds_off + 4 lane): for fp32 rows whose start is only dword-aligned
__device__ __forceinline__ void glds4bo(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}
// wave-uniform 64-bit pointer held in SGPRs
__device__ __forceinline__ const unsigned char* sgpr_ptr64(const void* p) {
    const uintptr_t v = (uintptr_t)p;
    return (const unsigned char*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
}
__device__ __forceinline__ uint4 lds_read16(unsigned addr) {      // ds_read_b128 at an LDS byte address
    typedef unsigned __attribute__((ext_vector_type(4))) u32x4_raw;
    const u32x4_raw v = *(const __attribute__((address_space(3))) u32x4_raw*)(uintptr_t)addr;
    return make_uint4(v[0], v[1], v[2], v[3]);
}
#define ST_DMA_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// counter-based dropout (DropCfg, launch.h): 32-bit murmur3-style mix of (seed, a, b) -- a, b = the two halves of the
// element's coordinates (attention: row = (item*H + head)*T + query, key; FFN: low / high word of the element index)
__host__ __device__ __forceinline__ unsigned drop_mix32(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// (see DropCfg in launch.h)  pair mix: one multiply + xor-shift of the two table entries
__host__ __device__ __forceinline__ unsigned drop_pair(unsigned rh, unsigned ch) {
    // rh and ch are mix32 outputs already: ONE odd multiply (carries break the XOR-linearity between the (row, column) pairs)
    // and one xor-shift that folds the well-mixed high product bits into the low half.  The attention kernels evaluate this for
    // every pair of probabilities, forward and three times backward: the two-multiply lowbias32 it replaces was ~8 of their
    // VALU instructions per element pair.  Statistics of the resulting masks (keep rate, neighbour / row / 2x2 correlations, row
    // and column sum variances) are indistinguishable from the old hash's.
    unsigned h = (rh ^ ch) * 0x9E3779B1u;
    return h ^ (h >> 15);
}
__host__ __device__ __forceinline__ unsigned drop_rowh(unsigned long long seed, unsigned row) { return drop_mix32((unsigned)seed ^ (row * 0x9E3779B1u)); }
__host__ __device__ __forceinline__ unsigned drop_colh(unsigned long long seed, unsigned j) { return drop_mix32((unsigned)(seed >> 32) ^ (j * 0x85ebca77u)); }
// the two keep factors (scale or 0) a 32-bit hash decides: .x for the even element of the pair, .y for the odd one
template <class D>
__device__ __forceinline__ float2 drop_factors2(const D& d, unsigned h) {
    return make_float2((h & 0xFFFFu) >= d.thresh16 ? d.scale : 0.0f, (h >> 16) >= d.thresh16 ? d.scale : 0.0f);
}
// FFN sites: hash of the element pair containing element index i (i even)
template <class D>
__device__ __forceinline__ unsigned drop_ffn_hash(const D& d, unsigned long long i) {
    return drop_mix32(((unsigned)d.seed ^ ((unsigned)(i >> 1) * 0x9E3779B1u)) + (unsigned)(d.seed >> 32));
}

// Fused-FFN weight stream (launch.h: launch_pack_ffn_stream; ffn_fused.h consumes it): element idx of one stage's
// F*256*3 values -> source offset in the fp32 conv weight and destination offset in the stream (16-bit elements).  hidden = 256
// hard-wired.  stage = 0: conv_1, 1: conv_2.
//   idx = ((c*24 + sl)*16 + f)*512 + lane*8 + e;  sl = (ci, tap, kp) = ci*6 + tap*2 + kp
//   f = ksl*8 + a8, row r = a8*32 + (lane&31), cin kk = ci*64 + (2kp+ksl)*16 + (lane>>5)*8 + e      (32x32x16 A fragments)
//   conv_1 (F, 256, 3): source (c*256 + r, kk, tap);  conv_2 (256, F, 3): source (r, c*256 + kk, tap)
__host__ __device__ __forceinline__ void ffn_stream_index(size_t idx, int stage, int F, size_t* src_off, size_t* dst_off) {
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), f = (int)((idx >> 9) & 15);
    const int sl = (int)((idx >> 13) % 24), c = (int)(idx / (24u * 8192u));
    const int ci = sl / 6, tap = (sl % 6) >> 1, kp = sl & 1;
    const int r = (f & 7) * 32 + (lane & 31), kk = ci * 64 + (2 * kp + (f >> 3)) * 16 + (lane >> 5) * 8 + e;
    if ((stage & 1) == 0) *src_off = ((size_t)(c * 256 + r) * 256 + kk) * 3 + tap;
    else                  *src_off = ((size_t)r * F + c * 256 + kk) * 3 + tap;
    *dst_off = (size_t)c * (48u * 8192u) + (size_t)((stage & 1) * 24 + sl) * 8192u + (idx & 8191);
}

// Weight stream of the Winograd F(2,3) fused FFN (ffn_wino.h): element idx of stage `stage` (0 = conv_1 (F, 256, 3), 1 = conv_2 (256, F, 3))
// -> offset of tap 0 of its (row, input channel) in the fp32 source, offset in the stream, plane (0: g0, 1: (g0 + g1 + g2) / 2, 2: g2).
// Stream = [chunk][stage][k-step 16][wave 8][plane 3] fragments of 512 elements, lane-linear: lane (l31, hi) holds row 32 wave + l31,
// k slots 16 k-step + 8 hi .. + 8.
__host__ __device__ __forceinline__ void ffn_wino_index(size_t idx, int stage, int F, size_t* src_off, size_t* dst_off, int* plane) {
    const int e = (int)(idx & 7), ln = (int)((idx >> 3) & 63);
    size_t t = idx >> 9;
    const int p = (int)(t % 3); t /= 3;
    const int w8 = (int)(t & 7), k = (int)((t >> 3) & 15), c = (int)(t >> 7);
    const int row = 32 * w8 + (ln & 31), kk = 16 * k + 8 * (ln >> 5) + e;
    *src_off = stage == 0 ? ((size_t)(c * 256 + row) * 256 + kk) * 3 : ((size_t)row * F + c * 256 + kk) * 3;
    *dst_off = (((((size_t)(c * 2 + stage) * 16 + k) * 8 + w8) * 3 + p) << 9) + (size_t)ln * 8 + e;
    *plane = p;
}
__host__ __device__ __forceinline__ float ffn_wino_plane(const float* g3, int plane) {
    return plane == 0 ? g3[0] : plane == 1 ? (g3[0] + g3[1] + g3[2]) * 0.5f : g3[2];
}

// Fragment-ordered copy of the fused q/k/v weight for qkv_ws.hip: element (output channel co of plane `plane`, input channel ci) ->
// [plane][wave = co / 32][k-step = ci / 16][lane = (ci % 16 / 8) * 32 + co % 32][ci % 8]: a wave's 16 MFMA fragments are 16
// consecutive 1-KiB pieces, lane-linear.
__host__ __device__ __forceinline__ size_t qkv_frag_index(int plane, int co, int ci) {
    return ((((size_t)(plane * 8 + (co >> 5)) * 16 + (ci >> 4)) * 64 + ((ci >> 3) & 1) * 32 + (co & 31)) << 3) + (ci & 7);
}

}  // namespace st
