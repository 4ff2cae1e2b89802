// LDS-DMA throughput microbenchmark (developer tool): how fast can one CU stream 1-KiB pieces global -> LDS, as a
// function of the source address pattern?  One 512-thread block per CU; every wave issues `iters` x 8 pieces and
// waits for them in groups of 8.  Patterns:
//   0: rows of 128 B with a large pitch (the conv kernels' operand tiles: 8 rows x 128 B per piece)
//   1: 1 KiB contiguous per piece (tile-packed operand)
//   2: like 0 but every CU reads the same 32 KiB (pure L2 hits)
//   3: like 1 but every CU reads the same 32 KiB
//   4: every CU walks the SAME 1.5 MiB weight matrix (256 rows x 6144 B) 128 B per row per stage: FFN conv_2's W stream
//   5: pattern 4 with the 1.5 MiB tile-packed (each stage's 32 KiB contiguous)
// build: hipcc --offload-arch=gfx950 -O3 dma_bench.hip -o dma_bench ; run: ./dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned char* lds_wave_base) {
    const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(off) : "memory");
}

template <int PATTERN>
__global__ __launch_bounds__(512, 1) void dma_kernel(const unsigned char* src, size_t per_block, int pitch, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 8 waves x 8 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool shared_src = PATTERN >= 2;
    if (PATTERN >= 4) {
        unsigned vo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = (wave * 4 + k) * 8 + (lane >> 3);
            vo[k] = PATTERN == 4 ? (unsigned)(row * 6144 + (lane & 7) * 16) : (unsigned)((wave * 4 + k) * 1024 + lane * 16);
        }
        for (int it = 0; it < iters * 2; ++it) {        // 32 KiB per stage
            const unsigned char* sb = src + (PATTERN == 4 ? (size_t)(it % 48) * 128 : (size_t)(it % 48) * 32768);
#pragma unroll
            for (int k = 0; k < 4; ++k) glds16s(sb, vo[k], smem + (wave * 4 + k) * 1024 + (it & 1) * 32768);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)(smem + 4 * (blockIdx.x & 63));
        return;
    }
    const unsigned char* base = src + (shared_src ? 0 : (size_t)blockIdx.x * per_block);
    unsigned voff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (PATTERN == 0 || PATTERN == 2) {      // piece = 8 rows x 128 B, row pitch `pitch`
            const int row = (wave * 8 + k) * 8 + (lane >> 3);
            voff[k] = (unsigned)(row * pitch + (lane & 7) * 16);
        } else {
            voff[k] = (unsigned)((wave * 8 + k) * 1024 + lane * 16);
        }
    }
    // stage stride: pattern 0 walks along the row (next 128 B of every row), pattern 1 the next 64-KiB slab
    const size_t step = (PATTERN == 0) ? 128 : (PATTERN == 1) ? 65536 : 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* sb = base + (size_t)it * step;
#pragma unroll
        for (int k = 0; k < 8; ++k) glds16s(sb, voff[k], smem + (wave * 8 + k) * 1024);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)(smem + 4 * (blockIdx.x & 63));
}

template <int PATTERN>
static void run(const char* name, const unsigned char* src, size_t per_block, int pitch, int iters, unsigned* sink) {
    hipFuncSetAttribute((const void*)dma_kernel<PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((dma_kernel<PATTERN>), dim3(256), dim3(512), 65536, 0, src, per_block, pitch, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const double bytes = 256.0 * iters * 65536.0;
        if (rep == 2) printf("%-34s %8.1f us  %7.2f TB/s  %6.1f B/us/CU = %.1f B/clk/CU @2.0GHz\n", name, ms * 1e3, bytes / ms / 1e9,
                             bytes / 256 / (ms * 1e3), bytes / 256 / (ms * 1e3) / 2000.0);
    }
}

int main() {
    const int iters = 48;                         // 48 x 64 KiB = 3 MiB per CU (one FFN conv_2 block's worth)
    const int pitch = 6144;                       // bytes between rows (k = 3 x 1024 channels x 2 B)
    const size_t per_block = (size_t)512 * pitch; // 512 rows per block
    const size_t total = 256 * per_block + (1 << 22);
    unsigned char* src; unsigned* sink;
    hipMalloc((void**)&src, total); hipMemset(src, 1, total); hipMalloc((void**)&sink, 4096);
    run<0>("strided rows, private per CU", src, per_block, pitch, iters, sink);
    run<1>("contiguous 1 KiB, private per CU", src, per_block, pitch, iters, sink);
    run<2>("strided rows, shared 32-64 KiB", src, per_block, pitch, iters, sink);
    run<3>("contiguous, shared 64 KiB", src, per_block, pitch, iters, sink);
    run<4>("shared 1.5 MiB W walk, strided", src, per_block, pitch, iters, sink);
    run<5>("shared 1.5 MiB W walk, packed", src, per_block, pitch, iters, sink);
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
