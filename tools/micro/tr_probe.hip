// Probe of ds_read_b64_tr_b16 (gfx950): every 16-bit LDS element holds its own index; each lane passes a byte address and
// prints the four 16-bit values it receives.  hipcc --offload-arch=gfx950 -O2 tools/micro/tr_probe.hip -o tools/micro/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(const int* addr_bytes, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (unsigned)addr_bytes[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    for (int test = 0; test < 3; ++test) {
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) {
            if (test == 0) addr[l] = l * 8;                 // lane l -> elements 4l .. 4l+3 (a dense 64 x 4 block)
            if (test == 1) addr[l] = (l & 15) * 128 + (l >> 4) * 8;    // 16 rows of 128 B (64 elements), lane group g at column block g
            if (test == 2) addr[l] = (l & 3) * 128 + ((l >> 2) & 3) * 8 + (l >> 4) * 512;
        }
        hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        std::vector<unsigned short> out(256);
        hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("test %d (lane: byte address -> 4 element indices received)\n", test);
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d addr %4d (elem %4d): %4d %4d %4d %4d\n", l, addr[l], addr[l] / 2, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    }
    return 0;
}
