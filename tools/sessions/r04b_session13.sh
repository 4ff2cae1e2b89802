#!/bin/bash
# marginal cost of each kernel class under the round-4 end-state default (two parts, weight-stationary q/k/v and out-projection)
ROOT=$(pwd); OUT=$ROOT/gpurun_out
{
for m in 0x10 0x20 0x40 0x200 0x400; do
  echo "== skip mask $m"; timeout 300 python tools/ab_engines.py "" "ST_SKIP_CLASSES=$m" 8 3 2>&1 | tail -1
done
} | tee $OUT/r04b_class_marginal_cost_final.txt
