// Is sustained f16 MFMA throughput on real data shape-dependent?  Pure register-resident MFMA loops (no LDS, no memory), every CU
// busy with 8 waves, operands = random / zero data: v_mfma_f32_32x32x16_f16 (8 accumulators of 16 registers) against
// v_mfma_f32_16x16x32_f16 (16 accumulators of 4 registers: 20 % less register-file traffic per FLOP).  Developer tool (round 4:
// the conv / FFN kernels are limited by the chip's power, not by their schedules -- DESIGN.md section 5).
// build: hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power ; run: ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void mfma_loop(const f16x8* src, int iters, float* sink) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(size_t)tid * 8 + i]; b[i] = src[(size_t)tid * 8 + 4 + i]; }
    float out = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) out += acc[i][r];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) out += acc[i][r];
    }
    sink[tid] = out;
}

int main() {
    const int blocks = 256 * 2, iters = 20000;
    const size_t n = (size_t)blocks * 512 * 8;
    std::vector<f16x8> h(n);
    f16x8* d; float* sink;
    hipMalloc(&d, n * sizeof(f16x8)); hipMalloc(&sink, (size_t)blocks * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fill = 0; fill < 3; ++fill) {
        unsigned s = 777u;
        for (auto& v : h) for (int k = 0; k < 8; ++k) {
            s = s * 1664525u + 1013904223u;
            const float u = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f;
            v[k] = (_Float16)(fill == 0 ? 0.f : fill == 1 ? u * 2.0f : u * 0.05f);
        }
        hipMemcpy(d, h.data(), n * sizeof(f16x8), hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            float ms32, ms16;
            hipEventRecord(e0); hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(512), 0, 0, d, iters, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms32, e0, e1);
            hipEventRecord(e0); hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(512), 0, 0, d, iters, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms16, e0, e1);
            // FLOPs: blocks * 8 waves * iters * (8 x 32768 | 16 x 16384)
            const double fl = (double)blocks * 8 * iters * 8 * 32768.0;
            printf("fill %s: 32x32x16 %.2f ms = %.0f TF/s   16x16x32 %.2f ms = %.0f TF/s\n", fill == 0 ? "zeros " : fill == 1 ? "U(-1,1)" : "small  ",
                   ms32, fl / ms32 * 1e-9, ms16, fl / ms16 * 1e-9);
        }
    }
    return 0;
}
