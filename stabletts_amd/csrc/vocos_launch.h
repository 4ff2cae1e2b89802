// Launcher interface of the Vocos vocoder kernels (vocos_kernels.hip); the GEMMs of the backbone and the head reuse
// the implicit-GEMM convolution kernels of launch.h.
// Reference: vocoders/vocos/models/backbone.py:50-56, module.py:33-46, head.py:39-72,93-117.
#pragma once
#include "launch.h"

namespace st {

constexpr int kVocDim = 512;        // backbone width the row kernels are built for (config.py:48)
constexpr int kVocNfft = 2048;      // config.py:6
constexpr int kVocHop = 512;        // config.py:8
constexpr int kVocBins = kVocNfft / 2 + 1;
constexpr int kVocHeadPlane = 1152; // channel pitch of the magnitude / phase planes in the head GEMM output (1025 -> 9 x 128)

// A16[b*T + t][j*M + c] = mel[b][c][t + j - 3]  (0 outside [0, T)): the k = 7 embed convolution as one GEMM with K = 7*M
hipError_t launch_voc_im2col7(int dtype, const float* mel, int B, int M, int T, void* a16, hipStream_t s);
// y = LayerNorm_512(x) * w + b  (eps 1e-6): out32 and/or out16 (may alias x for out32)
hipError_t launch_voc_ln(int dtype, const float* x, const float* w, const float* b, int64_t rows, float* out32, void* out16,
                         void* out16_lo, hipStream_t s);      // out16_lo (optional): x - float(round16(x)), the low half of a split-precision operand
// h16 = LayerNorm_512(dwconv7(x) + bias) * w + b   per utterance (rows of different items never mix); dw: [512][7]
hipError_t launch_voc_dwconv_ln(int dtype, const float* x, const float* dw, const float* dbias, const float* w, const float* b,
                                int B, int T, void* h16, hipStream_t s);
// head output rows [rows][2 * kVocHeadPlane] fp32 (log-magnitude plane, phase plane) -> windowed frames [rows][2048]
hipError_t launch_voc_spec_ifft(const float* head, const float* window, int64_t rows, float* frames, hipStream_t s);
// overlap-add with hop 512, "same" trim, envelope normalisation: audio[b][T*512]
hipError_t launch_voc_overlap_add(const float* frames, const float* window, int B, int T, float* audio, hipStream_t s);

}  // namespace st
