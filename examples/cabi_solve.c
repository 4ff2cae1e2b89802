/* Plain-C host of the C ABI (include/stabletts_hip.h): no Python, no torch, no C++.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/cabi_solve.c \
 *       -L stabletts_amd -lstabletts_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../stabletts_amd' -o examples/cabi_solve
 *   examples/cabi_solve [B] [T] [n_steps]
 *
 * Builds the 31M decoder configuration (config.py ModelConfig / MelConfig), discovers the expected tensors with
 * st_param_info, fills them with a seeded LCG (so that tests/test_gpu_parity.py can rebuild the same weights in
 * numpy), runs one Euler + CFG solve on device buffers and prints a checksum of the mel:
 *   cabi_solve B=.. T=.. n=.. sum=<double> abs=<double> first=<f> last=<f>
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stabletts_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define ST(x) do { int r_ = (x); if (r_ != ST_OK) { fprintf(stderr, "st error %d: %s (%s:%d)\n", r_, st_last_error(eng), __FILE__, __LINE__); return 3; } } while (0)

static uint32_t lcg_state;
static float lcg_uniform(void) {        /* U(-1, 1), 24-bit resolution */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (2.0f / 16777216.0f) - 1.0f;
}
static uint32_t name_seed(const char* s) {      /* FNV-1a of the tensor name: the stream of a tensor does not depend on order */
    uint32_t h = 2166136261u;
    for (; *s; ++s) { h ^= (uint8_t)*s; h *= 16777619u; }
    return h;
}
static float* dev_fill(size_t n, uint32_t seed, float scale) {
    float* h = (float*)malloc(n * sizeof(float));
    float* d = NULL;
    lcg_state = seed;
    for (size_t i = 0; i < n; ++i) h[i] = lcg_uniform() * scale;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    free(h);
    return d;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2, T = argc > 2 ? atoi(argv[2]) : 96, n_steps = argc > 3 ? atoi(argv[3]) : 4;
    const int M = 128, G = 256;
    st_config cfg = {M, 256, 1024, 4, 6, 3, G, ST_OPERAND_F16};   /* the shipping default: the parity-gated operand type */
    st_engine* eng = NULL;
    if (st_abi_version() != ST_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (st_create(&cfg, 0, &eng) != ST_OK) { fprintf(stderr, "st_create: %s\n", st_last_error(NULL)); return 1; }

    /* parameters: U(-1,1) * scale with scale = 1/sqrt(fan_in) (0.02 for the adaLN-Zero output layers) */
    const int np = st_num_params(eng);
    for (int i = 0; i < np; ++i) {
        const char* name; int64_t shape[4];
        const int nd = st_param_info(eng, i, &name, shape);
        if (nd < 1) { fprintf(stderr, "st_param_info failed\n"); return 1; }
        size_t n = 1; for (int k = 0; k < nd; ++k) n *= (size_t)shape[k];
        size_t fan_in = 1; for (int k = 1; k < nd; ++k) fan_in *= (size_t)shape[k];
        if (nd == 1) fan_in = (size_t)shape[0];
        float scale = 1.0f;
        for (size_t f = 1; f * f <= fan_in; ++f) scale = 1.0f / (float)f;      /* ~1/sqrt(fan_in), integer sqrt: portable */
        if (strstr(name, "adaLN_modulation.2")) scale = 0.02f;
        float* d = dev_fill(n, name_seed(name), scale);
        if (!d) { fprintf(stderr, "alloc failed\n"); return 2; }
        ST(st_load_param(eng, name, d, shape, nd));
        CK(hipFree(d));
    }
    ST(st_finalize(eng));

    /* inputs: (B, M, T) mu and z, (B, 1, T) mask with ragged lengths, (B, G) speakers, CFG null parameters */
    const size_t nbt = (size_t)B * M * T;
    float* mu = dev_fill(nbt, 11u, 1.0f);
    float* z = dev_fill(nbt, 12u, 1.0f);
    float* c = dev_fill((size_t)B * G, 13u, 1.0f);
    float* fs = dev_fill(G, 14u, 0.1f);
    float* fc = dev_fill(M, 15u, 0.1f);
    float* hmask = (float*)malloc((size_t)B * T * sizeof(float));
    for (int b = 0; b < B; ++b) {
        const int len = T - (b * T) / (3 * B);                      /* T, ..., down to ~2T/3 */
        for (int t = 0; t < T; ++t) hmask[(size_t)b * T + t] = t < len ? 1.0f : 0.0f;
    }
    float *mask = NULL, *out = NULL;
    CK(hipMalloc((void**)&mask, (size_t)B * T * sizeof(float)));
    CK(hipMemcpy(mask, hmask, (size_t)B * T * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&out, nbt * sizeof(float)));
    if (!mu || !z || !c || !fs || !fc) { fprintf(stderr, "alloc failed\n"); return 2; }

    hipStream_t stream; CK(hipStreamCreate(&stream));
    ST(st_cfm_solve(eng, mu, mask, z, c, n_steps, ST_SOLVER_EULER, 1, 3.0f, fs, fc, out, B, T, stream));
    CK(hipStreamSynchronize(stream));

    float* h = (float*)malloc(nbt * sizeof(float));
    CK(hipMemcpy(h, out, nbt * sizeof(float), hipMemcpyDeviceToHost));
    double sum = 0.0, asum = 0.0; int finite = 1;
    for (size_t i = 0; i < nbt; ++i) { sum += h[i]; asum += h[i] < 0 ? -h[i] : h[i]; if (!(h[i] == h[i]) || h[i] > 1e30f || h[i] < -1e30f) finite = 0; }
    printf("cabi_solve B=%d T=%d n=%d params=%d sum=%.9e abs=%.9e first=%.7e last=%.7e finite=%d\n", B, T, n_steps, np, sum, asum,
           (double)h[0], (double)h[nbt - 1], finite);
    st_destroy(eng);
    return finite ? 0 : 4;
}
