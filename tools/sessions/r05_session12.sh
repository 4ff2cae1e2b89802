#!/bin/bash
# Round 5, session 12: a phase offset between the two solve parts (the second part starts D us late)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for d in 30 60 95 130 160 250; do
  echo "== paired: no offset | second part $d us late"
  timeout 300 python tools/ab_engines.py "" "ST_PART_DELAY_US=$d" 8 3 2>&1 | tail -3
done
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_part_phase.txt
