#!/bin/bash
# builds (if needed) and runs the fused-FFN micro-benchmark; usage on the GPU box: bash tools/micro/run_ffn_bench.sh [N T]
cd "$(dirname "$0")"
[ -x ffn_bench ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value ffn_bench.hip -o ffn_bench || exit 1
./ffn_bench "$@"
