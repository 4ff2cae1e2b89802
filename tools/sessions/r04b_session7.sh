#!/bin/bash
# GPU call 7: ragged batches -- attention waves past the item's last needed frame skip their arithmetic
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "ragged or holey or mask or length" 2>&1 | tail -5 | tee $OUT/r04b_ragged_test.log
{
for i in 1 2 3; do
  for v in "ST_SPLIT=1 CLASS_TIMES_RAGGED=1" "CLASS_TIMES_RAGGED=1" "ST_SPLIT=1"; do
    echo -n "[$v] "; env $v timeout 300 python tools/class_times.py 2>&1 | tail -1 | cut -c1-200
  done
done
} | tee $OUT/r04b_ragged_attn_skip.txt
