#!/bin/bash
# Round 6, session 1: the new bench legs (ragged strong-scaling leg at N=1, 2-rank legs on the shared GPU), the new GPU tests, a
# fixed-count vs equal-cost bucket timing on real hardware, and the training-step kernel trace as the round's starting point.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python bench.py --steps 10 --warmup 2 2>$OUT/r06_s1_bench.err | tail -1 > $OUT/r06_s1_bench.json; cut -c1-300 $OUT/r06_s1_bench.json; tail -3 $OUT/r06_s1_bench.err
timeout 600 python bench.py --ragged --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/r06_s1_bench_ragged.json; cut -c1-300 $OUT/r06_s1_bench_ragged.json
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "bench or two_ranks" 2>&1 | tail -5
bash tools/profile_train.sh r06s1 > $OUT/r06_s1_profile_train.log 2>&1; tail -5 $OUT/r06_s1_profile_train.log
