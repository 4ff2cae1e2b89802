#!/bin/bash
# separate-process alternation (in-process A/B of different part counts shares HW queues between the two engines' streams)
ROOT=$(pwd); OUT=$ROOT/gpurun_out
{
for r in 0 1; do for i in 1 2 3 4; do
  for v in "ST_QKV_WS=0" "ST_SPLIT=2" "ST_SPLIT=1"; do
    echo -n "[ragged=$r $v] "; env CLASS_TIMES_RAGGED=$r $v timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-60
  done
done; done
} | tee $OUT/r04b_ab_parts_separate.txt
