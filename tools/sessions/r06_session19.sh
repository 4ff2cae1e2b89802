#!/bin/bash
# Round 6, session 19: training step -- cond-prenet SiLU in the GEMM epilogues (forward + dgrad), tiled batched per-item linears,
# first-round stagger of the phased k = 3 kernel (experiment), torch's fused AdamW (bench option); paired against the commit before.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{
timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -5
for i in 1 2 3; do
  STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06e.so tb before
  tb now
  ST_STAGGER_TICKS=500 tb stagger_5us
  ST_STAGGER_TICKS=1000 tb stagger_10us
  tb now_fused_adamw --fused-adamw
  ST_TRAIN_SIDE=0 tb now_no_side
done
} 2>&1 | tee $OUT/r06_s19_train.txt
