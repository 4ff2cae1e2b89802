// Micro-benchmark: the inner loop of the fused FFN's k = 3 convolution as (a) the shipped direct form and (b) Winograd F(2,3) along the
// frame axis with the input transform done ON THE FLY from raw rows and the fourth weight plane derived in registers
// ("A work reduction that the parity bar admits").  No LDS-DMA, no barriers: weights and activations sit in LDS,
// every wave loops over the same four k-steps -- it isolates what the matrix pipe, the LDS reads, the extra packed VALU work and the
// power limit make of "8 MFMAs + 10 reads + 32 packed adds" against "12 MFMAs + 12 reads" per k-step and wave.
//   block = 8 waves = 256 channels x 128 frames (wave: 64 channels x 64 frames / 32 frame pairs), one block per CU
//   direct : acc[2][2]    += W_tap[a] . H[rows l31 + tap (+32)]                         per k-step: 3 taps x (2 A + 2 B reads, 4 MFMAs)
//   F(2,3) : M[4][2]      += U_p[a] . V_p,  V = (d0-d2, d1+d2, d2-d1, d1-d3) of raw rows 2i .. 2i+3, U = (g0, (g0+g1+g2)/2, U0+U3-U1, g2)
//                                                                                       per k-step: 6 A + 4 B reads, 32 v_pk_add_f16, 8 MFMAs
//   raw rows for F(2,3) are stored pair-interleaved: row r -> storage row 2q + (e ^ (q & 1)), 16-byte slot c ^ ((q >> 1) & 7)
//   (q = r >> 1, e = r & 1): reads of rows 2i + e by lane i are conflict-free for ds_read_b128's lane groups.
// Also a correctness check of exactly this formulation against an fp64 convolution of the same 16-bit operands.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 wino_loop.hip -o wino_loop ; run: ./wino_loop [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../stabletts_amd/csrc/common.h"
using namespace st;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kRows = 136, kArea = kRows * 128;          // one 64-channel chunk of the activations
constexpr int kWStep = 24 * 1024;                        // weights of one k-step (16 input channels): 3 planes x 256 rows x 16 x 2 B
constexpr int kLds = kArea + 4 * kWStep;                 // 115,712 B

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

struct Frag { h2 v[4]; };
__device__ __forceinline__ Frag as_frag(uint4 u) { return __builtin_bit_cast(Frag, u); }
__device__ __forceinline__ f16x8_t as_v8(Frag f) { return __builtin_bit_cast(f16x8_t, f); }
__device__ __forceinline__ Frag fsub(Frag a, Frag b) { Frag r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
__device__ __forceinline__ Frag fadd(Frag a, Frag b) { Frag r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }

// MODE 0 = direct, 1 = F(2,3).  out: [256 channels][128 frames] fp32 (block 0 only).
template <int MODE>
__global__ __launch_bounds__(512, 1)
void loop_kernel(const unsigned char* img, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5, wc = wave >> 1, wf = wave & 1;
    for (int i = tid; i < kLds / 16; i += 512) ((uint4*)smem)[i] = ((const uint4*)img)[i];
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
    const unsigned wbase = lds0 + kArea + lane * 16 + wc * 2048;      // A fragment (plane p, channel fragment 2 wc + a) of k-step 0: + p * 8192 + a * 1024
    if constexpr (MODE == 0) {
        f32x16_t acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        unsigned badr[3][4];
        for (int j = 0; j < 3; ++j) for (int ks = 0; ks < 4; ++ks) {
            const int row = wf * 64 + l31 + j;
            badr[j][ks] = lds0 + row * 128 + (((ks * 2 + hi) ^ ((row >> 1) & 7)) << 4);
        }
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");      // the operands are re-read every iteration
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f16x8_t wfr[2], bfr[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) wfr[a] = as_vec8<OpF16>(lds_read16(wbase + ks * kWStep + j * 8192 + a * 1024));
#pragma unroll
                    for (int b = 0; b < 2; ++b) bfr[b] = as_vec8<OpF16>(lds_read16(badr[j][ks] + b * 4096));
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[a][b] = OpF16::mfma(wfr[a], bfr[b], acc[a][b]);
                }
        }
        if (blockIdx.x == 0)
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) {
                const int ch = wc * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, f = wf * 64 + b * 32 + l31;
                out[ch * 128 + f] = acc[a][b][r];
            }
        else asm volatile("" :: "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));
    } else if constexpr (MODE == 2) {
        // wave = 32 channels x 64 pairs (both frame halves): every A fragment belongs to ONE wave (the form a wave-private weight ring
        // needs); 64 more accumulator registers are kept live across the loop like the real kernel's conv_2 outputs
        f32x16_t M[4][2], Y[4];
        for (int p = 0; p < 4; ++p) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) M[p][b][r] = 0.f;
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) Y[k][r] = out[(k * 16 + r) * 64 + lane];
        asm volatile("" : "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]));
        const unsigned wb = lds0 + kArea + lane * 16 + wave * 1024;      // plane p: + p * 8192
        unsigned bbase[4];
        for (int e = 0; e < 4; ++e) {
            const int i = l31, q = i + (e >> 1), e1 = e & 1;
            const int srow = 2 * q + (e1 ^ (q & 1)), g = hi ^ ((q >> 1) & 7);
            bbase[e] = lds0 + srow * 128 + (g << 4);
        }
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const Frag U0 = as_frag(lds_read16(wb + ks * kWStep)), U1 = as_frag(lds_read16(wb + ks * kWStep + 8192)),
                           U3 = as_frag(lds_read16(wb + ks * kWStep + 16384));
                const Frag U2 = fsub(fadd(U0, U3), U1);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    Frag d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = as_frag(lds_read16((bbase[e] ^ (unsigned)(ks << 5)) + b * 8192));
                    const Frag V0 = fsub(d[0], d[2]), V1 = fadd(d[1], d[2]), V2 = fsub(d[2], d[1]), V3 = fsub(d[1], d[3]);
                    M[0][b] = OpF16::mfma(as_v8(U0), as_v8(V0), M[0][b]);
                    M[1][b] = OpF16::mfma(as_v8(U1), as_v8(V1), M[1][b]);
                    M[2][b] = OpF16::mfma(as_v8(U2), as_v8(V2), M[2][b]);
                    M[3][b] = OpF16::mfma(as_v8(U3), as_v8(V3), M[3][b]);
                }
            }
        }
        asm volatile("" : "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]));
        if (blockIdx.x == 0) {
            for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) {
                const int ch = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, f = 2 * (b * 32 + l31);
                out[ch * 128 + f] = M[0][b][r] + M[1][b][r] + M[2][b][r] + 0.f * Y[b][r];
                out[ch * 128 + f + 1] = M[1][b][r] - M[2][b][r] - M[3][b][r] + 0.f * Y[2 + b][r];
            }
        } else asm volatile("" :: "v"(M[0][0]), "v"(M[0][1]), "v"(M[1][0]), "v"(M[1][1]), "v"(M[2][0]), "v"(M[2][1]), "v"(M[3][0]), "v"(M[3][1]));
    } else {
        f32x16_t M[4][2];
        for (int p = 0; p < 4; ++p) for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) M[p][a][r] = 0.f;
        unsigned bbase[4];      // raw row 2 i + e of this lane's pair i, k-step 0 (k-step ks: ^ (ks << 5))
        for (int e = 0; e < 4; ++e) {
            const int i = wf * 32 + l31, q = i + (e >> 1), e1 = e & 1;
            const int srow = 2 * q + (e1 ^ (q & 1)), g = hi ^ ((q >> 1) & 7);
            bbase[e] = lds0 + srow * 128 + (g << 4);
        }
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                Frag U0[2], U1[2], U3[2], d[4];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    U0[a] = as_frag(lds_read16(wbase + ks * kWStep + 0 * 8192 + a * 1024));
                    U1[a] = as_frag(lds_read16(wbase + ks * kWStep + 1 * 8192 + a * 1024));
                    U3[a] = as_frag(lds_read16(wbase + ks * kWStep + 2 * 8192 + a * 1024));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = as_frag(lds_read16(bbase[e] ^ (unsigned)(ks << 5)));
                const Frag V0 = fsub(d[0], d[2]), V1 = fadd(d[1], d[2]), V2 = fsub(d[2], d[1]), V3 = fsub(d[1], d[3]);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const Frag U2 = fsub(fadd(U0[a], U3[a]), U1[a]);
                    M[0][a] = OpF16::mfma(as_v8(U0[a]), as_v8(V0), M[0][a]);
                    M[1][a] = OpF16::mfma(as_v8(U1[a]), as_v8(V1), M[1][a]);
                    M[2][a] = OpF16::mfma(as_v8(U2), as_v8(V2), M[2][a]);
                    M[3][a] = OpF16::mfma(as_v8(U3[a]), as_v8(V3), M[3][a]);
                }
            }
        }
        if (blockIdx.x == 0)
            for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) {
                const int ch = wc * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, f = 2 * (wf * 32 + l31);
                out[ch * 128 + f] = M[0][a][r] + M[1][a][r] + M[2][a][r];
                out[ch * 128 + f + 1] = M[1][a][r] - M[2][a][r] - M[3][a][r];
            }
        else asm volatile("" :: "v"(M[0][0]), "v"(M[0][1]), "v"(M[1][0]), "v"(M[1][1]), "v"(M[2][0]), "v"(M[2][1]), "v"(M[3][0]), "v"(M[3][1]));
    }
}

static unsigned rs = 777u;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    // operands: h[130 rows][64 ch] (16-bit), g[3 taps][256 ch][64 cin] (fp32 -> planes rounded to 16 bits)
    std::vector<_Float16> h(kRows * 64);
    std::vector<float> g(3 * 256 * 64);
    for (auto& v : h) v = (_Float16)((frand() + frand() + frand()) * 1.5f);
    for (auto& v : g) v = frand() * 0.1f;
    std::vector<unsigned char> img0(kLds, 0), img1(kLds, 0);
    auto put = [](std::vector<unsigned char>& im, size_t off, _Float16 v) { *(_Float16*)(im.data() + off) = v; };
    for (int r = 0; r < kRows; ++r) for (int k = 0; k < 64; ++k) {
        const int c = k >> 3, e8 = k & 7;
        put(img0, (size_t)r * 128 + ((c ^ ((r >> 1) & 7)) << 4) + e8 * 2, h[r * 64 + k]);
        const int q = r >> 1, e = r & 1, srow = 2 * q + (e ^ (q & 1));
        put(img1, (size_t)srow * 128 + ((c ^ ((q >> 1) & 7)) << 4) + e8 * 2, h[r * 64 + k]);
    }
    std::vector<_Float16> u0(256 * 64), u1(256 * 64), u3(256 * 64);
    for (int ch = 0; ch < 256; ++ch) for (int k = 0; k < 64; ++k) {
        const float g0 = g[(0 * 256 + ch) * 64 + k], g1 = g[(1 * 256 + ch) * 64 + k], g2 = g[(2 * 256 + ch) * 64 + k];
        u0[ch * 64 + k] = (_Float16)g0; u1[ch * 64 + k] = (_Float16)((g0 + g1 + g2) * 0.5f); u3[ch * 64 + k] = (_Float16)g2;
        for (int p = 0; p < 3; ++p) {
            const int ks = k >> 4, hi = (k >> 3) & 1, e8 = k & 7, cf = ch >> 5, l31 = ch & 31;
            const size_t off = (size_t)kArea + (size_t)ks * kWStep + (size_t)p * 8192 + cf * 1024 + (hi * 32 + l31) * 16 + e8 * 2;
            put(img0, off, (_Float16)g[(p * 256 + ch) * 64 + k]);
            put(img1, off, p == 0 ? u0[ch * 64 + k] : p == 1 ? u1[ch * 64 + k] : u3[ch * 64 + k]);
        }
    }
    // fp64 reference on the 16-bit direct operands
    std::vector<double> ref(256 * 128, 0.0);
    for (int ch = 0; ch < 256; ++ch) for (int f = 0; f < 128; ++f) {
        double s = 0;
        for (int j = 0; j < 3; ++j) for (int k = 0; k < 64; ++k) s += (double)(_Float16)g[(j * 256 + ch) * 64 + k] * (double)h[(f + j) * 64 + k];
        ref[ch * 128 + f] = s;
    }
    void *d0, *d1; float* dout;
    CK(hipMalloc(&d0, kLds)); CK(hipMalloc(&d1, kLds)); CK(hipMalloc(&dout, 256 * 128 * 4));
    CK(hipMemcpy(d0, img0.data(), kLds, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, img1.data(), kLds, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)loop_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipFuncSetAttribute((const void*)loop_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipFuncSetAttribute((const void*)loop_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    std::vector<float> o(256 * 128);
    double refmax = 0; for (double v : ref) refmax = fmax(refmax, fabs(v));
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(dout, 0, 256 * 128 * 4));
        if (mode == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(1), dim3(512), kLds, 0, (const unsigned char*)d0, dout, 1);
        else if (mode == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(1), dim3(512), kLds, 0, (const unsigned char*)d1, dout, 1);
        else hipLaunchKernelGGL(loop_kernel<2>, dim3(1), dim3(512), kLds, 0, (const unsigned char*)d1, dout, 1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
        double err = 0;
        const int fmax_ = 128;
        for (int ch = 0; ch < 256; ++ch) for (int f = 0; f < fmax_; ++f) err = fmax(err, fabs(o[ch * 128 + f] - ref[ch * 128 + f]));
        printf("%s: max |out - fp64 conv of the 16-bit operands| / max |ref| = %.3e\n", mode == 0 ? "direct" : mode == 1 ? "F(2,3)" : "F(2,3) 32 ch x 64 pairs", err / refmax);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned char> z(kLds, 0);
    void* dz; CK(hipMalloc(&dz, kLds)); CK(hipMemcpy(dz, z.data(), kLds, hipMemcpyHostToDevice));
    for (int data = 0; data < 2; ++data)
        for (int rep = 0; rep < 3; ++rep)
            for (int mode = 0; mode < 3; ++mode) {
                const unsigned char* src = data ? (const unsigned char*)dz : (const unsigned char*)(mode ? d1 : d0);
                float best = 1e30f;
                for (int t = 0; t < 5; ++t) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(256), dim3(512), kLds, 0, src, dout, iters);
                    else if (mode == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(256), dim3(512), kLds, 0, src, dout, iters);
                    else hipLaunchKernelGGL(loop_kernel<2>, dim3(256), dim3(512), kLds, 0, src, dout, iters);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms);
                }
                // algorithmic flops: 256 CUs x 256 ch x 128 frames x (iters x 64 cin) x 3 taps x 2
                const double fl = 256.0 * 256 * 128 * ((double)iters * 64) * 3 * 2;
                printf("%s data, %s: %8.1f us for %d x 4 k-steps  (%.0f algorithmic TFLOP/s)\n", data ? "zero  " : "random", mode == 0 ? "direct" : mode == 1 ? "F(2,3)" : "F(2,3) 32x64+Y",
                       best * 1e3, iters, fl / (best * 1e-3) * 1e-12);
            }
    return 0;
}
