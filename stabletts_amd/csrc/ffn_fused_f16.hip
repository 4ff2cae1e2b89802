#include "ffn_fused16.h"
namespace st {
hipError_t launch_ffn_fused_f16(const ConvGemmArgs& a, hipStream_t s) { return launch_ffn_fused_t<OpF16>(a, s); }
hipError_t launch_ffn_fused16_f16(const ConvGemmArgs& a, hipStream_t s) { return launch_ffn_fused16_t<OpF16>(a, s); }
}
