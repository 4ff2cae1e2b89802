#!/bin/bash
# Round 5, session 14: MORE (smaller) persistent blocks than CUs for the weight-stationary kernels in the two-part solve (separate processes, alternating)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for i in 1 2 3; do
  echo -n "[default 240/240] "; timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  echo -n "[qkv 384 blocks] "; ST_QKV_WS_BLOCKS=384 timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  echo -n "[qkv 768 blocks] "; ST_QKV_WS_BLOCKS=768 timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  echo -n "[oproj 512 blocks] "; ST_OPROJ_WS_BLOCKS=512 timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  echo -n "[oproj 1024 blocks] "; ST_OPROJ_WS_BLOCKS=1024 timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  echo -n "[both 768 / 1024] "; ST_QKV_WS_BLOCKS=768 ST_OPROJ_WS_BLOCKS=1024 timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
done
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_ws_more_blocks.txt
