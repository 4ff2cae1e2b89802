#include "conv_gemm_impl.h"
namespace st {
hipError_t launch_conv_gemm_f16(int taps, int epi, const ConvGemmArgs& a, hipStream_t s) {
    return launch_conv_gemm_t<OpF16>(taps, epi, a, s);
}
}  // namespace st
