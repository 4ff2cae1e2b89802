// Harness for stabletts_amd/csrc/ffn_wino.h: the Winograd F(2,3) fused FFN against the shipped fused kernel at the headline launch
// shape -- outputs compared with a tolerance (the formulation is not bit-identical: DESIGN.md section 7), both timed interleaved.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value ffn_wino_bench.hip -o ffn_wino_bench ; run: ./ffn_wino_bench [N T fill]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../stabletts_amd/csrc/ffn_wino.h"

using namespace st;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

template <int ABL>
static float run_ref(const ConvGemmArgs& a0, int reps, hipStream_t s) {
    CK(hipFuncSetAttribute((const void*)ffn_fused_kernel<OpF16, ABL, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds));
    ConvGemmArgs b = a0;
    b.tiles_f = (b.T + kFfnFusedFrames - 1) / kFfnFusedFrames; b.tiles_c = 1;
    const int total = b.n_items * b.tiles_f, grid = ((total + 7) / 8) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ffn_fused_kernel<OpF16, ABL, 0>), dim3(grid), dim3(512), kFfnLds, s, b);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ffn_fused_kernel<OpF16, ABL, 0>), dim3(grid), dim3(512), kFfnLds, s, b);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
    return ms * 1000.0f / reps;
}
template <int ABL>
static float run_wino(const ConvGemmArgs& a0, int reps, hipStream_t s) {
    CK(hipFuncSetAttribute((const void*)ffn_wino_kernel<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLds));
    ConvGemmArgs b = a0;
    b.tiles_f = (b.T + kFfnFusedFrames - 1) / kFfnFusedFrames; b.tiles_c = 1;
    const int total = b.n_items * b.tiles_f, grid = ((total + 7) / 8) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ffn_wino_kernel<ABL>), dim3(grid), dim3(512), kWnLds, s, b);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ffn_wino_kernel<ABL>), dim3(grid), dim3(512), kWnLds, s, b);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
    return ms * 1000.0f / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 1000, F = 1024, C = 256, reps = 20, rounds = 3;
    // data fill (argv[3]): 0 = uniform random (default), 1 = zeros (lowest switching power: how far is the kernel from its
    // schedule-bound time?), 2 = normal-like activations / small weights (closer to the model's statistics)
    const int fill = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t rows = (size_t)N * T;
    std::vector<_Float16> h(rows * C), w((size_t)2 * F * C * 3), w16((size_t)2 * F * C * 3);
    for (auto& v : h) v = (_Float16)(fill == 1 ? 0.0f : fill == 2 ? (frand() + frand() + frand() + frand()) * 1.7f : frand() * 2.0f);
    {   // conv_1 (F, 256, 3) and conv_2 (256, F, 3) weights, packed into the stream of either kernel (common.h: ffn_stream_index)
        std::vector<float> wsrc[2] = {std::vector<float>((size_t)F * C * 3), std::vector<float>((size_t)F * C * 3)};
        for (int st = 0; st < 2; ++st)
            for (auto& v : wsrc[st]) v = fill == 1 ? 0.0f : fill == 2 ? (frand() + frand() + frand()) * 0.04f : frand() * 0.08f;
        for (int st = 0; st < 2; ++st)
            for (size_t idx = 0; idx < (size_t)F * C * 3; ++idx) {
                size_t so, dof;
                ffn_stream_index(idx, st, F, &so, &dof);     w[dof] = (_Float16)wsrc[st][so];
                ffn_stream_index(idx, st | 2, F, &so, &dof); w16[dof] = (_Float16)wsrc[st][so];
            }
    }
    printf("fill mode %d\n", fill);
    std::vector<float> b1(F), b2(C), gate((size_t)N * C), mask((size_t)(N / 2) * T, 1.0f), x(rows * C), film(2 * C), ada((size_t)N * 2 * C);
    for (auto& v : b1) v = frand() * 0.1f;
    for (auto& v : b2) v = frand() * 0.1f;
    for (auto& v : gate) v = frand();
    for (auto& v : x) v = frand() * 2.0f;
    for (auto& v : film) v = 1.0f + frand() * 0.1f;
    for (auto& v : ada) v = frand() * 0.1f;
    void *dh, *dw, *dw16, *db1, *db2, *dgate, *dmask, *dx, *dxo, *do16, *dln, *dfilm, *dada, *dz;
    CK(hipMalloc(&dh, h.size() * 2)); CK(hipMalloc(&dw, w.size() * 2)); CK(hipMalloc(&dw16, w16.size() * 2)); CK(hipMalloc(&db1, F * 4)); CK(hipMalloc(&db2, C * 4));
    CK(hipMalloc(&dgate, gate.size() * 4)); CK(hipMalloc(&dmask, mask.size() * 4)); CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dxo, x.size() * 4));
    CK(hipMalloc(&do16, rows * C * 2)); CK(hipMalloc(&dln, rows * C * 2)); CK(hipMalloc(&dfilm, film.size() * 4)); CK(hipMalloc(&dada, ada.size() * 4));
    CK(hipMalloc(&dz, 256)); CK(hipMemset(dz, 0, 256));
    CK(hipMemcpy(dh, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1, b1.data(), F * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgate, gate.data(), gate.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dmask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfilm, film.data(), film.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dada, ada.data(), ada.size() * 4, hipMemcpyHostToDevice));

    ConvGemmArgs a; memset(&a, 0, sizeof(a));
    a.a0 = dh; a.c0 = C; a.a0_mod = N; a.a1_mod = N; a.w = dw; a.bias = (const float*)db2; a.bias1 = (const float*)db1; a.cmid = F;
    a.cout = C; a.T = T; a.n_items = N; a.mask = (const float*)dmask; a.mask_mod = N / 2; a.flags = GF_SILU | GF_MASK;
    a.gate = (const float*)dgate; a.gate_stride = C; a.res32 = (const float*)dx; a.out32 = (float*)dxo; a.out16 = do16; a.zeros = dz;
    a.ln_h16 = dln; a.ln_film = (const float*)dfilm; a.ln_film_stride = 0; a.ln_film_mod = 1;
    a.ln_ada = (const float*)dada; a.ln_ada_stride = 2 * C; a.ln_shift_off = 0; a.ln_scale_off = C; a.ln_mask_out = 0;

    // the Winograd stream: [chunk][stage][k-step][wave][plane U0, U1, U3] fragments of 1 KiB, lane-linear (lane (l31, hi): row 32 w + l31,
    // k slots 16 k + 8 hi .. + 8); planes from the fp32 taps, rounded to f16 once
    std::vector<_Float16> ww((size_t)2 * F * C * 3);
    {
        rng_state = 12345u;      // the same weights as above: replay the generator up to the weight draws
        for (size_t i = 0; i < h.size(); ++i) { if (fill == 2) { frand(); frand(); frand(); frand(); } else if (fill != 1) frand(); }
        std::vector<float> wsrc[2] = {std::vector<float>((size_t)F * C * 3), std::vector<float>((size_t)F * C * 3)};
        for (int st = 0; st < 2; ++st)
            for (auto& v : wsrc[st]) v = fill == 1 ? 0.0f : fill == 2 ? (frand() + frand() + frand()) * 0.04f : frand() * 0.08f;
        const int nch = F / 256;
        for (int c = 0; c < nch; ++c) for (int st = 0; st < 2; ++st) for (int k = 0; k < 16; ++k) for (int w8 = 0; w8 < 8; ++w8)
            for (int ln = 0; ln < 64; ++ln) for (int e = 0; e < 8; ++e) {
                const int row = 32 * w8 + (ln & 31), kk = 16 * k + 8 * (ln >> 5) + e;
                float g3[3];
                for (int j = 0; j < 3; ++j)
                    g3[j] = st == 0 ? wsrc[0][((size_t)(c * 256 + row) * 256 + kk) * 3 + j] : wsrc[1][((size_t)row * F + c * 256 + kk) * 3 + j];
                const float u[3] = {g3[0], (g3[0] + g3[1] + g3[2]) * 0.5f, g3[2]};
                const size_t base = ((((size_t)(c * 2 + st) * 16 + k) * 8 + w8) * 3) * 512 + (size_t)ln * 8 + e;
                for (int p = 0; p < 3; ++p) ww[base + (size_t)p * 512] = (_Float16)u[p];
            }
    }
    void* dww; CK(hipMalloc(&dww, ww.size() * 2)); CK(hipMemcpy(dww, ww.data(), ww.size() * 2, hipMemcpyHostToDevice));
    ConvGemmArgs aw = a; aw.w = dww;
    const double gflop = 2.0 * 2.0 * rows * (double)F * C * 3 * 1e-9;
    std::vector<unsigned short> o16(rows * C);
    std::vector<float> ref16(rows * C), refx(rows * C), ox(rows * C);
    auto f16f = [](unsigned short u) { _Float16 hh; memcpy(&hh, &u, 2); return (float)hh; };
    run_ref<0>(a, 1, s); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o16.data(), dln, o16.size() * 2, hipMemcpyDeviceToHost)); for (size_t i = 0; i < o16.size(); ++i) ref16[i] = f16f(o16[i]);
    CK(hipMemcpy(refx.data(), dxo, refx.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dln, 0, o16.size() * 2)); CK(hipMemset(dxo, 0, ox.size() * 4));
    run_wino<0>(aw, 1, s); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o16.data(), dln, o16.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ox.data(), dxo, ox.size() * 4, hipMemcpyDeviceToHost));
    {
        double e16 = 0, m16 = 0, ex = 0, mx = 0; size_t bad = 0, nan = 0;
        for (size_t i = 0; i < o16.size(); ++i) {
            const float v = f16f(o16[i]);
            if (!(v == v) || !(ox[i] == ox[i])) { ++nan; continue; }
            e16 = fmax(e16, fabs(v - ref16[i])); m16 = fmax(m16, fabs(ref16[i]));
            ex = fmax(ex, fabs(ox[i] - refx[i])); mx = fmax(mx, fabs(refx[i]));
            bad += fabs(ox[i] - refx[i]) > 0.02 * mx + 1e-3;
        }
        printf("F(2,3) vs shipped kernel: LN output max |diff| %.3e (max |ref| %.3f), residual stream max |diff| %.3e (max |ref| %.3f), > 2 %%: %zu, NaN: %zu\n",
               e16, m16, ex, mx, bad, nan);
        if (bad || nan) {
            int shown = 0;
            for (size_t i = 0; i < ox.size() && shown < 12; ++i)
                if (!(ox[i] == ox[i]) || fabs(ox[i] - refx[i]) > 0.02 * mx + 1e-3) {
                    printf("   item %zu frame %zu ch %zu: %g vs %g\n", i / C / T, (i / C) % T, i % C, ox[i], refx[i]); ++shown;
                }
        }
    }
    // run-to-run determinism (a protocol race shows here)
    { std::vector<float> ox2(ox.size()); run_wino<0>(aw, 1, s); CK(hipDeviceSynchronize()); CK(hipMemcpy(ox2.data(), dxo, ox2.size() * 4, hipMemcpyDeviceToHost));
      size_t d = 0; for (size_t i = 0; i < ox.size(); ++i) d += memcmp(&ox[i], &ox2[i], 4) != 0; printf("rerun: %zu of %zu outputs differ\n", d, ox.size()); }
    for (int r = 0; r < rounds; ++r) {
        { const float us = run_ref<0>(a, reps, s);   printf("round %d  shipped      : %7.1f us  %6.0f TF/s\n", r, us, gflop / us * 1e3); }
        { const float us = run_wino<0>(aw, reps, s); printf("round %d  F(2,3)       : %7.1f us  %6.0f TF/s (algorithmic)\n", r, us, gflop / us * 1e3); }
        { const float us = run_ref<1>(a, reps, s);   printf("round %d  shipped, no epilogue : %7.1f us\n", r, us); }
        { const float us = run_wino<1>(aw, reps, s); printf("round %d  F(2,3), no epilogue  : %7.1f us\n", r, us); }
        if (r == 2) {
            { const float us = run_wino<3>(aw, reps, s);  printf("         F(2,3) no epilogue, no DMA in the steps : %7.1f us\n", us); }
            { const float us = run_wino<5>(aw, reps, s);  printf("         F(2,3) no epilogue, no barriers         : %7.1f us\n", us); }
            { const float us = run_wino<9>(aw, reps, s);  printf("         F(2,3) no epilogue, no transform step   : %7.1f us\n", us); }
            { const float us = run_wino<17>(aw, reps, s); printf("         F(2,3) no epilogue, no MFMAs            : %7.1f us\n", us); }
            { const float us = run_wino<33>(aw, reps, s); printf("         F(2,3) no epilogue, no top waits        : %7.1f us\n", us); }
            { const float us = run_wino<15>(aw, reps, s); printf("         F(2,3) steps only (no DMA, barriers, transform) : %7.1f us\n", us); }
        }
        fflush(stdout);
    }
    return 0;
}
