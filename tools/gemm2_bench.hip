// Micro-benchmark of conv_gemm2 (LDS-staged epilogue, row-complete tiles) against the first-generation
// LDS-DMA kernel (developer tool).  The fp32-staged epilogues (EPI_F32 / EPI_RESGATE without LayerNorm) must agree bitwise
// with the old kernel; EPI_ACT16 starts its accumulators from the bias, so it agrees to rounding only ("MISMATCH" is expected).
#include "../stabletts_amd/csrc/conv_gemm_impl.h"
#include "../stabletts_amd/csrc/conv_gemm2_impl.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace st;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint32_t rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static void* dev_bf16(size_t n, float scale) {
    std::vector<uint16_t> h(n); for (size_t i = 0; i < n; ++i) h[i] = f2bf(frand() * scale);
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d;
}
static float* dev_f32(size_t n, float scale, float offset = 0.f) {
    std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = frand() * scale + offset;
    float* d; CK(hipMalloc((void**)&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}
static std::vector<uint8_t> fetch(const void* d, size_t bytes) {
    std::vector<uint8_t> h(bytes); CK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost)); return h;
}

template <int TAPS, int EPI>
static void run_shape(const char* name, int items, int T, int c0, int c1, int cout, int reps) {
    const int cin = c0 + c1;
    const size_t rows = (size_t)items * T;
    ConvGemmArgs a; memset(&a, 0, sizeof(a));
    a.a0 = dev_bf16(rows * c0, 1.0f); a.c0 = c0;
    if (c1) { a.a1 = dev_bf16(rows * c1, 1.0f); a.c1 = c1; }
    a.a0_mod = items; a.a1_mod = items;
    a.w = dev_bf16((size_t)cout * TAPS * cin, 0.05f);
    a.bias = dev_f32(cout, 0.1f);
    a.cout = cout; a.T = T; a.n_items = items;
    a.tiles_f = (T + kBF - 1) / kBF; a.tiles_c = cout / kBC;
    a.mask = dev_f32(rows, 0.f, 1.0f); a.mask_mod = items;
    a.flags = getenv("GB_FLAGS") ? atoi(getenv("GB_FLAGS")) : (GF_SILU | GF_MASK);
    const int Tp = (T + 63) / 64 * 64;
    const size_t out16_bytes = rows * cout * 2, out32_bytes = rows * cout * 4;
    CK(hipMalloc(&a.out16, out16_bytes)); CK(hipMalloc((void**)&a.out32, out32_bytes));
    a.gate = dev_f32((size_t)items * cout, 0.2f); a.gate_stride = cout;
    size_t vt_bytes = 0;
    if (EPI == EPI_QKV) {
        const int C = cout / 3;
        CK(hipMalloc(&a.q, rows * C * 2)); CK(hipMalloc(&a.k, rows * C * 2));
        vt_bytes = (size_t)items * C * Tp * 2;
        CK(hipMalloc(&a.vt, vt_bytes));
        a.rope_cos = dev_f32((size_t)T * 16, 1.0f); a.rope_sin = dev_f32((size_t)T * 16, 1.0f);
        a.Tp = Tp; a.qscale = 0.18f; a.n_heads = C / 64;
    }
    { void* z; CK(hipMalloc(&z, 256)); CK(hipMemset(z, 0, 256)); a.zeros = z; }
    const double flops = 2.0 * rows * cout * (double)cin * TAPS;
    auto snapshot = [&](auto launch) {
        CK(hipMemset(a.out16, 0, out16_bytes)); CK(hipMemset(a.out32, 0, out32_bytes));
        if (EPI == EPI_QKV) { CK(hipMemset(a.q, 0, rows * (cout / 3) * 2)); CK(hipMemset(a.k, 0, rows * (cout / 3) * 2)); CK(hipMemset(a.vt, 0, vt_bytes)); }
        CK(launch()); CK(hipDeviceSynchronize());
        std::vector<std::vector<uint8_t>> r;
        if (EPI == EPI_QKV) { r.push_back(fetch(a.q, rows * (cout / 3) * 2)); r.push_back(fetch(a.k, rows * (cout / 3) * 2)); r.push_back(fetch(a.vt, vt_bytes)); }
        else { r.push_back(fetch(a.out16, out16_bytes)); r.push_back(fetch(a.out32, out32_bytes)); }
        return r;
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            for (int i = 0; i < 2; ++i) CK(launch());
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) CK(launch());
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms / reps);
        }
        return best;
    };
    auto l_old = [&] { return launch_glds<OpBF16, TAPS, EPI, 0>(a, nullptr); };
    auto l_t128 = [&] { return launch_g2<OpBF16, TAPS, EPI, 128, 128, 2, 2>(a, nullptr); };
    auto l_rc = [&] { return launch_g2<OpBF16, TAPS, EPI, 256, 128, 4, 2>(a, nullptr); };
    const auto ref = snapshot(l_old);
    auto cmp = [&](const std::vector<std::vector<uint8_t>>& x) {
        bool same = (x == ref);
        if (!same && EPI == EPI_QKV) {       // FMA contraction may differ: report the bf16 max abs difference
            for (size_t k = 0; k < x.size(); ++k) {
                double md = 0; size_t nd = 0;
                const uint16_t* p = (const uint16_t*)x[k].data(); const uint16_t* q = (const uint16_t*)ref[k].data();
                for (size_t i = 0; i < x[k].size() / 2; ++i) {
                    uint32_t a = (uint32_t)p[i] << 16, b = (uint32_t)q[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
                    if (p[i] != q[i]) { ++nd; md = fmax(md, fabs((double)fa - fb)); }
                }
                printf("   [%s plane %zu: %zu of %zu differ, max abs diff %.4g]\n", name, k, nd, x[k].size() / 2, md);
            }
        }
        return same;
    };
    const bool ok1 = cmp(snapshot(l_t128));
    const bool ok2 = (cout % 256 == 0) ? cmp(snapshot(l_rc)) : true;
    printf("%-6s K=%4dx%d N=%4d  old %7.1f us %6.1f TF/s |", name, cin, TAPS, cout, 0.f, 0.f);
    const float t_old = timeit(l_old), t1 = timeit(l_t128);
    printf("\r%-6s K=%4dx%d N=%4d  old %7.1f us %6.1f TF/s | g2-128x128 %7.1f us %6.1f TF/s %s", name, cin, TAPS, cout,
           t_old * 1e3, flops / (t_old * 1e-3) / 1e12, t1 * 1e3, flops / (t1 * 1e-3) / 1e12, ok1 ? "==" : "MISMATCH");
    if (cout % 256 == 0) {
        const float t2 = timeit(l_rc);
        printf(" | g2-256x128 %7.1f us %6.1f TF/s %s", t2 * 1e3, flops / (t2 * 1e-3) / 1e12, ok2 ? "==" : "MISMATCH");
        if (cout == 256 && EPI != EPI_ACT16) {      // + fused FiLM/LayerNorm/modulate epilogue
            ConvGemmArgs b = a;
            CK(hipMalloc(&b.ln_h16, rows * 256 * 2));
            b.ln_film = dev_f32(512, 0.5f, 1.0f); b.ln_film_stride = 0; b.ln_film_mod = 1;
            b.ln_ada = dev_f32((size_t)items * 1536, 0.1f); b.ln_ada_stride = 1536; b.ln_shift_off = 0; b.ln_scale_off = 256;
            const float t3 = timeit([&] { return launch_g2<OpBF16, TAPS, EPI, 256, 128, 4, 2>(b, nullptr); });
            printf(" | +LN %7.1f us", t3 * 1e3);
        }
    }
#if ST_STAGE_TIMING
    {
        unsigned long long* dbg; CK(hipMalloc((void**)&dbg, 64 * 2 * 8 * 8)); CK(hipMemset(dbg, 0, 64 * 2 * 8 * 8));
        ConvGemmArgs b = a; b.dbg = dbg;
        auto run = [&](const char* tag, auto launch) {
            CK(hipMemset(dbg, 0, 64 * 2 * 8 * 8)); CK(launch()); CK(launch()); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(64 * 2 * 8); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            double s[7] = {0}; int nb = 0;
            for (int i = 0; i < 128; ++i) if (h[i * 8 + 7]) { for (int k = 0; k < 7; ++k) s[k] += (double)h[i * 8 + k]; ++nb; }
            if (!nb) return;
            const double n = s[4] / nb;
            printf("\n   [%s timing, %d waves sampled, %.0f stages: per stage (memtime ticks) issue %.0f | ds_read+mfma %.0f | dma wait %.0f | barrier %.0f ; loop total %.0f, epilogue %.0f]",
                   tag, nb, n, s[0] / s[4], s[1] / s[4], s[2] / s[4], s[3] / s[4], s[5] / nb, s[6] / nb);
        };
        run("128x128", [&] { return launch_g2<OpBF16, TAPS, EPI, 128, 128, 2, 2>(b, nullptr); });
        if (cout % 256 == 0) run("256x256", [&] { return launch_g2<OpBF16, TAPS, EPI, 256, 256, 2, 4>(b, nullptr); });
    }
#endif
    if (cout % 256 == 0) {
        auto l_y = [&] { return launch_g2<OpBF16, TAPS, EPI, 256, 256, 2, 4>(a, nullptr); };
        const bool oky = cmp(snapshot(l_y));
        const float ty = timeit(l_y);
        printf(" | g2-256x256(2x4) %7.1f us %6.1f TF/s %s", ty * 1e3, flops / (ty * 1e-3) / 1e12, oky ? "==" : "MISMATCH");
        auto l_y2 = [&] { return launch_g2<OpBF16, TAPS, EPI, 256, 256, 4, 2>(a, nullptr); };
        const bool oky2 = cmp(snapshot(l_y2));
        const float ty2 = timeit(l_y2);
        printf(" | g2-256x256(4x2) %7.1f us %6.1f TF/s %s", ty2 * 1e3, flops / (ty2 * 1e-3) / 1e12, oky2 ? "==" : "MISMATCH");
    }
    if constexpr (TAPS == 3) {
        auto l_g3 = [&] { return launch_g3<OpBF16, EPI>(a, nullptr); };
        const bool ok3 = cmp(snapshot(l_g3));
        const float t4 = timeit(l_g3);
        printf(" | g3(3-buf) %7.1f us %6.1f TF/s %s", t4 * 1e3, flops / (t4 * 1e-3) / 1e12, ok3 ? "==" : "MISMATCH");
        if (cout % 256 == 0) {
            auto l_g3b = [&] { return launch_g3<OpBF16, EPI, 256, 256, 2, 4>(a, nullptr); };
            const bool ok3b = cmp(snapshot(l_g3b));
            const float t5 = timeit(l_g3b);
            printf(" | g3-256x256 %7.1f us %6.1f TF/s %s", t5 * 1e3, flops / (t5 * 1e-3) / 1e12, ok3b ? "==" : "MISMATCH");
        }
    }
    printf("\n"); fflush(stdout);
}

int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 64;
    const int T = argc > 2 ? atoi(argv[2]) : 1000;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    printf("items %d, T %d\n", items, T);
    run_shape<3, EPI_ACT16>("ffn1", items, T, 256, 0, 1024, reps);
    if (getenv("GB_ONLY_FFN1")) return 0;
    run_shape<3, EPI_RESGATE>("ffn2", items, T, 1024, 0, 256, reps);
    run_shape<3, EPI_F32>("lsc", items, T, 256, 256, 256, reps);
    run_shape<1, EPI_RESGATE>("oproj", items, T, 256, 0, 256, reps);
    run_shape<1, EPI_F32>("final", items, T, 256, 0, 128, reps);
    return 0;
}
