// Stand-alone benchmark of the weight-stationary q/k/v projection (stabletts_amd/csrc/qkv_ws.hip) at the headline launch shape
// (64 CFG-doubled items x 1000 frames) and at one solve part's (16 items); ablation variants (ST_QKV_WS_VAR bits) interleaved.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 qkv_bench.hip -o qkv_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>
#include "../../stabletts_amd/csrc/qkv_ws.hip"
using namespace st;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static unsigned rs = 12345u;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 1000, C = 256, H = 4, Tp = (T + 63) / 64 * 64, NMAX = 64;
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<_Float16> h((size_t)NMAX * T * C), w((size_t)768 * C);
    for (auto& v : h) v = (_Float16)((frand() + frand() + frand() + frand()) * 1.7f);
    for (auto& v : w) v = (_Float16)(frand() * 0.12f);
    std::vector<float> bias(768), rc((size_t)T * 16), rsn((size_t)T * 16);
    for (auto& v : bias) v = frand() * 0.1f;
    for (size_t i = 0; i < rc.size(); ++i) { rc[i] = cosf(0.001f * i); rsn[i] = sinf(0.001f * i); }
    std::vector<_Float16> wfr(w.size());
    for (int co = 0; co < 768; ++co) for (int ci = 0; ci < C; ++ci) wfr[qkv_frag_index(co / 256, co % 256, ci)] = w[(size_t)co * C + ci];
    void *dh, *dw, *db, *dq, *dk, *dv, *dz, *dsink, *drc, *drs;
    CK(hipMalloc(&dh, h.size() * 2)); CK(hipMalloc(&dw, w.size() * 2)); CK(hipMalloc(&db, 768 * 4));
    CK(hipMalloc(&dq, (size_t)NMAX * T * C * 2)); CK(hipMalloc(&dk, (size_t)NMAX * T * C * 2)); CK(hipMalloc(&dv, (size_t)NMAX * Tp * C * 2));
    CK(hipMalloc(&dz, 256)); CK(hipMemset(dz, 0, 256)); CK(hipMalloc(&dsink, 65536)); CK(hipMalloc(&drc, rc.size() * 4)); CK(hipMalloc(&drs, rc.size() * 4));
    CK(hipMemcpy(dh, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, wfr.data(), wfr.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bias.data(), 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drs, rsn.data(), rc.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int N : {64, 32, 16}) {
        ConvGemmArgs a; memset(&a, 0, sizeof(a));
        a.a0 = dh; a.c0 = C; a.a0_mod = N; a.w = dw; a.w_frag = dw; a.bias = (const float*)db; a.cout = 768; a.T = T; a.n_items = N;
        a.q = dq; a.k = dk; a.vt = dv; a.rope_cos = (const float*)drc; a.rope_sin = (const float*)drs; a.Tp = Tp; a.qscale = 0.18f; a.n_heads = H;
        a.zeros = dz; a.sink = dsink;
        const int tiles_f = (T + 63) / 64;
        for (int Lopt : {0}) {
            int L = 85 / tiles_f; if (L < 1) L = 1;
            if (Lopt == 1) L = 170 / tiles_f;          // two blocks' worth of lists per CU (blocks queue: 1 per CU fits)
            if (Lopt == 2) L = 42 / tiles_f > 0 ? 42 / tiles_f : 1;
            if (L > N) L = N;
            const int grid = ((3 * tiles_f * L + 7) / 8) * 8;
            auto run = [&](auto tag) {
                constexpr int VAR = decltype(tag)::value;
                CK(hipFuncSetAttribute((const void*)qkv_ws_kernel<OpF16, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, kQwsLds));
                const int reps = 20;
                hipLaunchKernelGGL((qkv_ws_kernel<OpF16, VAR>), dim3(grid), dim3(512), kQwsLds, s, a, L);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((qkv_ws_kernel<OpF16, VAR>), dim3(grid), dim3(512), kQwsLds, s, a, L);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
                printf("N %2d L %2d grid %3d (%.1f tiles per block) var %2d : %6.1f us\n", N, L, grid, (double)N * tiles_f / (tiles_f * L), VAR, ms * 1000 / reps);
            };
            run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 2>{}); run(std::integral_constant<int, 4>{});
            run(std::integral_constant<int, 6>{}); run(std::integral_constant<int, 14>{}); run(std::integral_constant<int, 22>{});
            run(std::integral_constant<int, 30>{}); run(std::integral_constant<int, 38>{});
        }
    }
    {   // loop anatomy (var 64: s_memtime stamps; ticks of the 100 MHz constant-rate counter -> ns)
        ConvGemmArgs a; memset(&a, 0, sizeof(a));
        const int N = 64, tiles_f = (T + 63) / 64; int L = 85 / tiles_f; if (L < 1) L = 1;
        a.a0 = dh; a.c0 = C; a.a0_mod = N; a.w = dw; a.w_frag = dw; a.bias = (const float*)db; a.cout = 768; a.T = T; a.n_items = N;
        a.q = dq; a.k = dk; a.vt = dv; a.rope_cos = (const float*)drc; a.rope_sin = (const float*)drs; a.Tp = Tp; a.qscale = 0.18f; a.n_heads = H;
        a.zeros = dz; a.sink = dsink;
        unsigned long long* dd; CK(hipMalloc(&dd, 64 * 2 * 8 * 8)); CK(hipMemset(dd, 0, 64 * 2 * 8 * 8)); a.dbg = dd;
        const int grid = ((3 * tiles_f * L + 7) / 8) * 8;
        CK(hipFuncSetAttribute((const void*)qkv_ws_kernel<OpF16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, kQwsLds));
        hipLaunchKernelGGL((qkv_ws_kernel<OpF16, 64>), dim3(grid), dim3(512), kQwsLds, s, a, L);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> hd(64 * 2 * 8); CK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
        const char* nm[8] = {"top wait", "barrier A", "DMA issue", "reads+MFMA", "epilogue", "barrier B", "stores", "prologue"};
        printf("anatomy (s_memtime ticks / 100, totals over the block's tiles; blocks 0 / 1 / 2 / 8, wave 0 | wave 4):\n");
        for (int k = 0; k < 8; ++k) {
            printf("  %-12s", nm[k]);
            for (int blk : {0, 1, 2, 8}) printf("  %8.2f | %8.2f ", hd[(blk * 2 + 0) * 8 + k] / 100.0, hd[(blk * 2 + 1) * 8 + k] / 100.0);
            printf("\n");
        }
    }
    return 0;
}
