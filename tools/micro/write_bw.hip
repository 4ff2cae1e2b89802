// HBM write / read / copy bandwidth of MI355X under the store forms the engine uses (developer micro-benchmark, round 5).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/write_bw tools/micro/write_bw.hip && tools/micro/write_bw
// Every block streams its own contiguous slice in 1-KiB wave rows (64 lanes x 16 B), like the row epilogues.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 = plain store, 1 = sc1 (write-through, the engine's row stores), 2 = nt, 3 = sc0 sc1, 4 = read (sum), 5 = copy (plain), 6 = copy (sc1 stores), 7 = copy (nt loads, sc1 stores)
__global__ __launch_bounds__(512) void bw_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, unsigned* sink) {
    const size_t per_block = n16 / gridDim.x;
    uint4* d = dst + (size_t)blockIdx.x * per_block;
    const uint4* s = src + (size_t)blockIdx.x * per_block;
    uint4 acc = make_uint4(threadIdx.x, 1, 2, 3);
    for (size_t i = threadIdx.x; i < per_block; i += 512) {
        if constexpr (MODE >= 4) {
            uint4 v;
            if constexpr (MODE == 7) { typedef unsigned nt4 __attribute__((ext_vector_type(4))); nt4 t = __builtin_nontemporal_load((const nt4*)(s + i)); v = make_uint4(t.x, t.y, t.z, t.w); }
            else v = s[i];
            if constexpr (MODE == 4) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; continue; }
            acc = v;
        }
        const u32x4_t vv = {acc.x, acc.y, acc.z, acc.w};
        if constexpr (MODE == 0 || MODE == 5) d[i] = acc;
        else if constexpr (MODE == 1 || MODE == 6 || MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d + i), "v"(vv) : "memory");
        else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(d + i), "v"(vv) : "memory");
        else if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(d + i), "v"(vv) : "memory");
    }
    if constexpr (MODE == 4) if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

template <int MODE> void run(const char* name, uint4* a, uint4* b, size_t bytes, int blocks, unsigned* sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((bw_kernel<MODE>), dim3(blocks), dim3(512), 0, 0, a, b, n16, sink);
    CK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((bw_kernel<MODE>), dim3(blocks), dim3(512), 0, 0, a, b, n16, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
    const double moved = (MODE >= 5 ? 2.0 : 1.0) * bytes;
    printf("%-34s %5d blocks  %7.1f MB : %7.1f us  %6.2f TB/s%s\n", name, blocks, bytes / 1e6, ms * 1000 / reps, moved / (ms / reps * 1e-3) / 1e12, MODE >= 5 ? " (read + write)" : "");
}

int main() {
    const size_t cap = 2048ull << 20;
    uint4 *a, *b; unsigned* sink;
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, cap)); CK(hipMemset(b, 2, cap));
    for (size_t mb : {100ull, 1024ull}) {          // 100 MB: what one glue-kernel launch writes (fits the 256-MB Infinity Cache); 1 GB: past it
        const size_t bytes = mb << 20;
        for (int blocks : {256, 1024}) {
            run<0>("write, plain stores", a, b, bytes, blocks, sink);
            run<1>("write, sc1 (write-through) stores", a, b, bytes, blocks, sink);
            run<2>("write, nt stores", a, b, bytes, blocks, sink);
            run<3>("write, sc0 sc1 stores", a, b, bytes, blocks, sink);
            run<4>("read", a, b, bytes, blocks, sink);
            run<5>("copy, plain", a, b, bytes, blocks, sink);
            run<6>("copy, sc1 stores", a, b, bytes, blocks, sink);
            run<7>("copy, nt loads + sc1 stores", a, b, bytes, blocks, sink);
        }
    }
    return 0;
}
