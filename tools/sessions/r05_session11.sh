#!/bin/bash
# Round 5, session 11: three / four solve parts WITH the weight-stationary kernels (their launch-size thresholds lowered)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for i in 1 2; do
  echo -n "[ST_SPLIT=2 default thresholds] "; ST_SPLIT=2 timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  for sp in 3 4; do
    echo -n "[ST_SPLIT=$sp default thresholds] "; ST_SPLIT=$sp timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
    echo -n "[ST_SPLIT=$sp ws from 200 / 400 tiles] "; ST_SPLIT=$sp ST_QKV_WS_MIN_TILES=200 ST_OPROJ_WS_MIN_TILES=400 timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  done
done
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_parts_ws_thresholds.txt
