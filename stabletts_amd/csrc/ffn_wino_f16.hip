#include "ffn_wino.h"
namespace st {
hipError_t launch_ffn_wino_f16(const ConvGemmArgs& a, hipStream_t s) { return launch_ffn_wino_impl(a, s); }
}
