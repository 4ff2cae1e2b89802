#!/bin/bash
# GPU call 9: solve parts x q/k/v kernel, paired against the current default (4 parts, generic tile); all-ones and ragged
ROOT=$(pwd); OUT=$ROOT/gpurun_out
{
for r in 0 1; do
  for v in "ST_SPLIT=2 ST_QKV_WS=1" "ST_SPLIT=3 ST_QKV_WS=1" "ST_SPLIT=2 ST_QKV_WS=0" "ST_SPLIT=4 ST_QKV_WS=1 ST_QKV_WS_MIN_TILES=2000"; do
    echo "== ragged=$r  default vs [$v]"; AB_RAGGED=$r timeout 300 python tools/ab_engines.py "" "$v" 10 3 2>&1 | tail -3
  done
done
} | tee $OUT/r04b_ab_parts_qkv_ws.txt
