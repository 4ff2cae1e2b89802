#include "conv_gemm2_inst.h"
namespace st {
hipError_t launch_conv_gemm2_bf16(int cfg, int taps, int epi, const ConvGemmArgs& a, hipStream_t s) {
    return launch_conv_gemm2_t<OpBF16>(cfg, taps, epi, a, s);
}
}  // namespace st
