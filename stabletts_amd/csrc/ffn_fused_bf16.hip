#include "ffn_fused.h"
namespace st {
hipError_t launch_ffn_fused_bf16(const ConvGemmArgs& a, hipStream_t s) { return launch_ffn_fused_t<OpBF16>(a, s); }
}
