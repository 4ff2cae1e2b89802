#!/bin/bash
# Round 6, session 21: weight-gradient GEMMs in order on the main stream, only their plane reductions on the side stream
# (ST_TRAIN_SIDE=3, two plane buffers) against the default (GEMM + reduction on the side stream) and no side streams.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{
ST_TEST_SIDE_MODE=3 timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k bitwise_neutral 2>&1 | tail -3
for i in 1 2 3; do
  tb side_default
  ST_TRAIN_SIDE=3 tb gemm_main_reduce_side
  ST_TRAIN_SIDE=0 tb no_side
done
} 2>&1 | tee $OUT/r06_s21_train.txt
