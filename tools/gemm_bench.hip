// Standalone micro-benchmark of the implicit-GEMM conv kernel variants on the hot-path shapes
// (developer tool; build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -o tools/gemm_bench).
// Times each (shape, variant) with HIP events and checks that variants agree bitwise
// (identical accumulation order).  Random full-range bf16 data (guide rule 25: never zero-fill).
#include "../stabletts_amd/csrc/conv_gemm_impl.h"

#include <cstdio>
#include <cstring>
#include <vector>

using namespace st;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint32_t rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

static void* dev_bf16(size_t n, float scale) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(frand() * scale);
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static float* dev_f32(size_t n, float scale, float offset = 0.f) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = frand() * scale + offset;
    float* d; CK(hipMalloc((void**)&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <int TAPS, int EPI, int VAR>
static float time_variant(const ConvGemmArgs& a, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) CK((launch_var<OpBF16, TAPS, EPI, VAR>(a, nullptr)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) CK((launch_var<OpBF16, TAPS, EPI, VAR>(a, nullptr)));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
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
    a.flags = GF_SILU | GF_MASK;
    const int Tp = (T + 63) / 64 * 64;
    size_t out16_bytes = rows * cout * 2, out32_bytes = rows * cout * 4;
    CK(hipMalloc(&a.out16, out16_bytes));
    CK(hipMalloc((void**)&a.out32, out32_bytes));
    a.gate = dev_f32((size_t)items * cout, 0.2f); a.gate_stride = cout;
    if (EPI == EPI_QKV) {
        const int C = cout / 3;
        CK(hipMalloc(&a.q, rows * C * 2)); CK(hipMalloc(&a.k, rows * C * 2));
        CK(hipMalloc(&a.vt, (size_t)items * C * Tp * 2));
        a.rope_cos = dev_f32((size_t)T * 16, 1.0f); a.rope_sin = dev_f32((size_t)T * 16, 1.0f);
        a.Tp = Tp; a.qscale = 0.18f; a.n_heads = C / 64;
    }
    { void* z; CK(hipMalloc(&z, 256)); CK(hipMemset(z, 0, 256)); a.zeros = z; }
    const double flops = 2.0 * rows * cout * (double)cin * TAPS;
    // correctness: LDS-DMA kernel vs the register-staged reference variant (bitwise: same accumulation order)
    CK(hipMemset(a.out16, 0, out16_bytes)); CK(hipMemset(a.out32, 0, out32_bytes));
    CK((launch_var<OpBF16, TAPS, EPI, 0>(a, nullptr))); CK(hipDeviceSynchronize());
    auto r16 = fetch(EPI == EPI_QKV ? a.q : a.out16, EPI == EPI_QKV ? rows * (cout / 3) * 2 : out16_bytes);
    auto r32 = fetch(a.out32, out32_bytes);
    CK(hipMemset(a.out16, 0, out16_bytes)); CK(hipMemset(a.out32, 0, out32_bytes));
    CK((launch_glds<OpBF16, TAPS, EPI, 0>(a, nullptr))); CK(hipDeviceSynchronize());
    auto g16 = fetch(EPI == EPI_QKV ? a.q : a.out16, EPI == EPI_QKV ? rows * (cout / 3) * 2 : out16_bytes);
    auto g32 = fetch(a.out32, out32_bytes);
    const bool same = (r16 == g16) && (r32 == g32);
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
    struct { const char* note; float ms; } res[] = {
        {"glds (shipping)", timeit([&] { return launch_glds<OpBF16, TAPS, EPI, 0>(a, nullptr); })},
        {"glds + setprio", timeit([&] { return launch_glds<OpBF16, TAPS, EPI, 1>(a, nullptr); })},
        {"ablate: compute only (no LDS-DMA)", timeit([&] { return launch_glds<OpBF16, TAPS, EPI, 2>(a, nullptr); })},
        {"ablate: LDS-DMA + ds_read only (no MFMA)", timeit([&] { return launch_glds<OpBF16, TAPS, EPI, 4>(a, nullptr); })},
        {"exp: 1 A buffer, 3 blocks/CU (racy)", timeit([&] { return launch_glds<OpBF16, TAPS, EPI, 8>(a, nullptr); })},
        {"exp: 3 blocks/CU + setprio (racy)", timeit([&] { return launch_glds<OpBF16, TAPS, EPI, 9>(a, nullptr); })},
        {"register-staged (old)", timeit([&] { return launch_var<OpBF16, TAPS, EPI, 0>(a, nullptr); })},
    };
    for (auto& r : res)
        printf("%-6s K=%4dx%d N=%4d  %8.1f us  %7.1f TF/s  %s%s\n", name, cin, TAPS, cout, r.ms * 1e3,
               flops / (r.ms * 1e-3) / 1e12, r.note, (&r == &res[0]) ? (same ? "  [bitwise == old]" : "  [MISMATCH]") : "");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 64;
    const int T = argc > 2 ? atoi(argv[2]) : 1000;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, CUs %d, items %d, T %d\n", prop.gcnArchName, prop.multiProcessorCount, items, T);
    run_shape<3, EPI_ACT16>("ffn1", items, T, 256, 0, 1024, reps);
    run_shape<3, EPI_RESGATE>("ffn2", items, T, 1024, 0, 256, reps);
    run_shape<3, EPI_F32>("lsc", items, T, 256, 256, 256, reps);
    if (argc > 4) {
        run_shape<1, EPI_QKV>("qkv", items, T, 256, 0, 768, reps);
        run_shape<1, EPI_RESGATE>("oproj", items, T, 256, 0, 256, reps);
        run_shape<1, EPI_F32>("final", items, T, 256, 0, 128, reps);
    }
    return 0;
}
