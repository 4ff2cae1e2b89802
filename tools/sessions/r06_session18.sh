#!/bin/bash
# Round 6, session 18: Vocos with split weights in the ConvNeXt blocks' pointwise convs: parity (tests/test_gpu_vocos.py, printed errors) and cost.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
{ timeout 900 python -m pytest tests/test_gpu_vocos.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -25
timeout 300 python tools/vocos_bench.py 2>&1 | grep -v amdgpu.ids | tail -6; } | tee $OUT/r06_s18_vocos_split_weights.txt
