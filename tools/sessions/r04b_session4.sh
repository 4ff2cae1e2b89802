#!/bin/bash
# GPU call 4: out projection on 256 x 128 row-complete tiles (ST_OPROJ_RC=1) vs 256 x 256, all-ones and ragged batches
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
{
for i in 1 2; do
  for v in "ST_OPROJ_RC=0" "ST_OPROJ_RC=1" "ST_OPROJ_RC=0 CLASS_TIMES_RAGGED=1" "ST_OPROJ_RC=1 CLASS_TIMES_RAGGED=1" "ST_OPROJ_RC=0 ST_SPLIT=1" "ST_OPROJ_RC=1 ST_SPLIT=1" "ST_OPROJ_RC=0 ST_SPLIT=1 CLASS_TIMES_RAGGED=1" "ST_OPROJ_RC=1 ST_SPLIT=1 CLASS_TIMES_RAGGED=1"; do
    echo -n "[$v] "; env $v timeout 300 python tools/class_times.py 2>&1 | tail -1
  done
done
} | tee $OUT/r04b_ab_oproj_rc.txt
