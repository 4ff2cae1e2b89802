#include "conv_gemm2_inst.h"
namespace st {
hipError_t launch_conv_gemm2_bf16(int cfg, int taps, int epi, const ConvGemmArgs& a, hipStream_t s) {
    return launch_conv_gemm2_t<OpBF16>(cfg, taps, epi, a, s);
}
hipError_t launch_splitk_finish_bf16(int epi, const ConvGemmArgs& a, const float* part, int S, hipStream_t s) {
    return launch_splitk_finish_t<OpBF16>(epi, a, part, S, s);
}
}  // namespace st
