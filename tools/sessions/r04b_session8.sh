#!/bin/bash
# GPU call 8: marginal cost of each kernel class in the default four-part solve and in the single-sequence solve (launches of
# the class skipped; paired, interleaved; profile class ids: QKV 4, ATTN 5, OPROJ 6, FFN2 9, LSC 10)
ROOT=$(pwd); OUT=$ROOT/gpurun_out
{
for m in 0x10 0x20 0x40 0x200 0x400; do
  echo "== skip mask $m (default parts)"; timeout 300 python tools/ab_engines.py "" "ST_SKIP_CLASSES=$m" 12 3 2>&1 | tail -3
  echo "== skip mask $m (ST_SPLIT=1)"; ST_SPLIT=1 timeout 300 python tools/ab_engines.py "" "ST_SKIP_CLASSES=$m" 8 3 2>&1 | tail -3
done
} | tee $OUT/r04b_class_marginal_cost.txt
