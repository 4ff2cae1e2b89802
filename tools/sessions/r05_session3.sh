#!/bin/bash
# Round 5, session 3: balanced work lists of the weight-stationary kernels (qkv_ws unit ranges, oproj_ws 256 blocks)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "weight_stationary or ragged_tile or two_part" 2>&1 | tail -4 | tee $OUT/r05_ws_tests.log
BASE=$ROOT/tools/ab/base.so
{
for r in 0 1; do
  echo "== ragged=$r, one launch sequence (ST_SPLIT=1): base | new"
  for i in 1 2; do
    echo -n "[base] "; env CLASS_TIMES_RAGGED=$r ST_SPLIT=1 STABLETTS_HIP_LIB=$BASE timeout 200 python tools/class_times.py 2>&1 | tail -1
    echo -n "[new ] "; env CLASS_TIMES_RAGGED=$r ST_SPLIT=1 timeout 200 python tools/class_times.py 2>&1 | tail -1
  done
done
echo "== paired, default parts: base | new"
timeout 400 python tools/ab_engines.py "STABLETTS_HIP_LIB=$BASE" "" 16 3 2>&1 | tail -4
echo "== paired ragged"
AB_RAGGED=1 timeout 400 python tools/ab_engines.py "STABLETTS_HIP_LIB=$BASE" "" 12 3 2>&1 | tail -4
echo "== paired: oproj 224 blocks | 256 blocks (new lib both)"
timeout 400 python tools/ab_engines.py "ST_OPROJ_WS_BLOCKS=240" "" 12 3 2>&1 | tail -4
echo "== paired ST_SPLIT=1: base | new"
ST_SPLIT=1 timeout 400 python tools/ab_engines.py "STABLETTS_HIP_LIB=$BASE" "" 8 3 2>&1 | tail -4
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_ws_balance.txt
