#include "conv_gemm2_inst.h"
namespace st {
hipError_t launch_conv_gemm2_f16(int cfg, int taps, int epi, const ConvGemmArgs& a, hipStream_t s) {
    return launch_conv_gemm2_t<OpF16>(cfg, taps, epi, a, s);
}
hipError_t launch_splitk_finish_f16(int epi, const ConvGemmArgs& a, const float* part, int S, hipStream_t s) {
    return launch_splitk_finish_t<OpF16>(epi, a, part, S, s);
}
}  // namespace st
