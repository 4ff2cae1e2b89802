#!/bin/bash
# GPU call 3: number of solve parts (launch sequences on separate streams): 4 (default) vs 6 / 8, eager and HIP graph
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
{
for i in 1 2; do
  for v in "ST_SPLIT=4" "ST_SPLIT=6" "ST_SPLIT=8" "ST_SPLIT=8 ST_HIP_GRAPH=1" "ST_SPLIT=4 ST_HIP_GRAPH=1"; do
    echo -n "[$v] "; env $v timeout 300 python tools/class_times.py 2>&1 | tail -1 | cut -c1-60
  done
done
} | tee $OUT/r04b_ab_parts.txt
