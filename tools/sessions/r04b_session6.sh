#!/bin/bash
# GPU call 6: qkv_ws experiments (ST_QKV_WS_VAR: 1 plain stores, 2 no stores, 4 no LDS-DMA after the first two tiles)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
{
for v in "ST_QKV_WS=0" "ST_QKV_WS=1" "ST_QKV_WS=1 ST_QKV_WS_VAR=1" "ST_QKV_WS=1 ST_QKV_WS_VAR=2" "ST_QKV_WS=1 ST_QKV_WS_VAR=4" "ST_QKV_WS=1 ST_QKV_WS_VAR=6"; do
    echo -n "[$v] "; env $v ST_SPLIT=1 timeout 300 python tools/class_times.py 2>&1 | tail -1 | cut -c1-130
done
} | tee $OUT/r04b_qkv_ws_var.txt
