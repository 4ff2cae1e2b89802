#!/bin/bash
# Round 6, session 17: the T = 1000 training-trajectory test.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu -s -k "trajectory" 2>&1 | grep -E "T=1000|lr=|passed|failed|Error|assert" | tee $OUT/r06_s17_trajectory_T1000.txt
