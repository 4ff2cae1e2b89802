#!/bin/bash
# Round 6, session 20: (a) what dropout costs the training step (FFN epilogue hash, attention masks); (b) the K split of the
# weight-gradient GEMM (blocks per launch) for the cond prenet's 1024 x 1024 x 3 conv alone and for every GEMM (experiment knob).
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{
for i in 1 2 3; do
  tb now
  tb no_dropout --no-dropout
  ST_XP_TARGET_TN=3016 tb big_128
  ST_XP_TARGET_TN=3064 tb big_512
  ST_XP_TARGET_TN=3096 tb big_768
  ST_XP_TARGET_TN=16 tb all_128
  ST_XP_TARGET_TN=64 tb all_512
done
} 2>&1 | tee $OUT/r06_s20_train.txt
