// Issue cost of the integer instructions the dropout pair hash uses (drop_pair, csrc/common.h): dependent chains of v_mul_lo_u32,
// v_mul_u32_u24, v_xor_b32 + v_lshrrev, v_add_u32 per wave, one wave per SIMD and four, timed with s_memtime (100 MHz) against the
// shader clock (wall_clock64 not needed: ratios between the chains are what matters).  Developer tool:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/imul_rate.hip -o tools/micro/imul_rate && tools/micro/imul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void chain(unsigned* out, unsigned long long* ticks, int iters) {
    unsigned x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x5bd1e995u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (KIND == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));
            else if (KIND == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y));
            else if (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
            else if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
            else if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned* out; unsigned long long* ticks;
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&ticks, 64 * 8));
    const char* names[6] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_add_u32", "v_exp_f32", "v_fma_f32"};
    for (int waves = 1; waves <= 4; waves *= 4) {      // one block on one CU: 4 SIMDs x `waves` waves per SIMD
        for (int kind = 0; kind < 6; ++kind) {
            const int iters = 2000, threads = 64 * 4 * waves;
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(chain<0>, dim3(1), dim3(threads), 0, 0, out, ticks, iters); break;
                    case 1: hipLaunchKernelGGL(chain<1>, dim3(1), dim3(threads), 0, 0, out, ticks, iters); break;
                    case 2: hipLaunchKernelGGL(chain<2>, dim3(1), dim3(threads), 0, 0, out, ticks, iters); break;
                    case 3: hipLaunchKernelGGL(chain<3>, dim3(1), dim3(threads), 0, 0, out, ticks, iters); break;
                    case 4: hipLaunchKernelGGL(chain<4>, dim3(1), dim3(threads), 0, 0, out, ticks, iters); break;
                    default: hipLaunchKernelGGL(chain<5>, dim3(1), dim3(threads), 0, 0, out, ticks, iters); break;
                }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
            }
            printf("%d wave(s) per SIMD  %-14s  %8.2f ns per instruction per wave (100-MHz ticks x 10 / %d)\n", waves, names[kind], h * 10.0 / (iters * 32.0), iters * 32);
        }
    }
    return 0;
}
