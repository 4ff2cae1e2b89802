#!/bin/bash
# GPU call 5: weight-stationary q/k/v projection (qkv_ws.hip): parity test, per-class times with / without it
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -s -k "weight_stationary" 2>&1 | tail -15 | tee $OUT/r04b_qkv_ws_test.log
{
for i in 1 2; do
  for v in "ST_QKV_WS=0 ST_SPLIT=1" "ST_QKV_WS=1 ST_SPLIT=1" "ST_QKV_WS=0" "ST_QKV_WS=1" "ST_QKV_WS=0 CLASS_TIMES_RAGGED=1" "ST_QKV_WS=1 CLASS_TIMES_RAGGED=1"; do
    echo -n "[$v] "; env $v timeout 300 python tools/class_times.py 2>&1 | tail -1
  done
done
} | tee $OUT/r04b_ab_qkv_ws.txt
