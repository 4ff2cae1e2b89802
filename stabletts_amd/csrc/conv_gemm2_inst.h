// Instantiation list of the second-generation conv GEMM for one operand type (included by the two
// conv_gemm2_<dtype>.hip translation units).
#pragma once
#include "conv_gemm2_impl.h"
#include "conv_gemm_phased.h"

namespace st {

// cfg 0: T128 (128 ch x 128 frames, 4 waves)   cfg 1: RC (256 ch x 128 frames, 8 waves, LayerNorm- and QKV-capable)
template <class P>
static hipError_t launch_conv_gemm2_t(int cfg, int taps, int epi, const ConvGemmArgs& a, hipStream_t s) {
    if (cfg == 0) {
        if (taps == 3 && epi == EPI_ACT16) return launch_g2<P, 3, EPI_ACT16, 128, 128, 2, 2>(a, s);
        if (taps == 1 && epi == EPI_F32) return launch_g2<P, 1, EPI_F32, 128, 128, 2, 2>(a, s);
        if (taps == 3 && epi == EPI_F32) return launch_g2<P, 3, EPI_F32, 128, 128, 2, 2>(a, s);
        if (taps == 1 && epi == EPI_RESGATE) return launch_g2<P, 1, EPI_RESGATE, 128, 128, 2, 2>(a, s);
        if (taps == 3 && epi == EPI_RESGATE) return launch_g2<P, 3, EPI_RESGATE, 128, 128, 2, 2>(a, s);
        if (taps == 1 && epi == EPI_GELU16) return launch_g2<P, 1, EPI_GELU16, 128, 128, 2, 2>(a, s);
    } else if (cfg == 1) {
        if (taps == 3 && epi == EPI_F32) return launch_g2<P, 3, EPI_F32, 256, 128, 4, 2>(a, s);
        if (taps == 1 && epi == EPI_F32) return launch_g2<P, 1, EPI_F32, 256, 128, 4, 2>(a, s);
        if (taps == 3 && epi == EPI_RESGATE) return launch_g2<P, 3, EPI_RESGATE, 256, 128, 4, 2>(a, s);
        if (taps == 1 && epi == EPI_RESGATE) return launch_g2<P, 1, EPI_RESGATE, 256, 128, 4, 2>(a, s);
        if (taps == 1 && epi == EPI_QKV) return launch_g2<P, 1, EPI_QKV, 256, 128, 4, 2>(a, s);
    } else if (cfg == 3) {   // 256 ch x 256 frames, 8 waves (2x4) of 128x64: least LDS traffic per MFMA, LayerNorm-capable
        if (taps == 3 && epi == EPI_ACT16) return launch_g2<P, 3, EPI_ACT16, 256, 256, 2, 4>(a, s);
        if (taps == 3 && epi == EPI_F32) return launch_g2<P, 3, EPI_F32, 256, 256, 2, 4>(a, s);
        if (taps == 1 && epi == EPI_F32) return launch_g2<P, 1, EPI_F32, 256, 256, 2, 4>(a, s);
        if (taps == 3 && epi == EPI_RESGATE) return launch_g2<P, 3, EPI_RESGATE, 256, 256, 2, 4>(a, s);
        if (taps == 1 && epi == EPI_RESGATE) return launch_g2<P, 1, EPI_RESGATE, 256, 256, 2, 4>(a, s);
        if (taps == 1 && epi == EPI_QKV) return launch_g2<P, 1, EPI_QKV, 256, 256, 2, 4>(a, s);
        if (taps == 1 && epi == EPI_GELU16) return launch_g2<P, 1, EPI_GELU16, 256, 256, 2, 4>(a, s);
    } else if (cfg == 4) {   // T64: 128 ch x 64 frames, 4 waves of 64x32 -- twice the blocks of T128 for latency-bound small grids
        if (taps == 3 && epi == EPI_ACT16) return launch_g2<P, 3, EPI_ACT16, 128, 64, 2, 2>(a, s);
        if (taps == 1 && epi == EPI_F32) return launch_g2<P, 1, EPI_F32, 128, 64, 2, 2>(a, s);
        if (taps == 3 && epi == EPI_F32) return launch_g2<P, 3, EPI_F32, 128, 64, 2, 2>(a, s);
    } else if (cfg == 5) {   // RC64: 256 ch x 64 frames, 8 waves of 64x32 (QKV planes of small grids)
        if (taps == 1 && epi == EPI_QKV) return launch_g2<P, 1, EPI_QKV, 256, 64, 4, 2>(a, s);
    } else if (cfg == 6) {   // 256 x 254 tile, k = 3, phased K loop with three weight buffers (conv_gemm_phased.h)
        if (taps == 3 && epi == EPI_ACT16) return launch_phased3<P, EPI_ACT16>(a, s);
        if (taps == 3 && epi == EPI_F32) return launch_phased3<P, EPI_F32>(a, s);
        if (taps == 3 && epi == EPI_RESGATE) return launch_phased3<P, EPI_RESGATE>(a, s);
        if (taps == 3 && epi == EPI_SILU) return launch_phased3<P, EPI_SILU>(a, s);      // training FFN (the only tile that carries it)
    } else if (cfg == 7) {   // RC1: 256 ch x 128 frames with ONE weight buffer -> 74 KB, two resident blocks (QKV on big grids)
        if (taps == 1 && epi == EPI_QKV) return launch_g2<P, 1, EPI_QKV, 256, 128, 4, 2, 1>(a, s);
        // (EPI_RESGATE + fused LayerNorm -- out-proj -- does not fit 128 VGPRs on this tile: 54 spilled registers)
    } else if (cfg == 2) {   // 3-buffer k=3 kernel (counted vmcnt), 128 ch x 126 frames
        if (taps == 3 && epi == EPI_ACT16) return launch_g3<P, EPI_ACT16>(a, s);
        if (taps == 3 && epi == EPI_F32) return launch_g3<P, EPI_F32>(a, s);
        if (taps == 3 && epi == EPI_RESGATE) return launch_g3<P, EPI_RESGATE>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace st
