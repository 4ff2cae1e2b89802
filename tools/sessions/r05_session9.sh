#!/bin/bash
# Round 5, session 9: s_setprio around the MFMA clusters, kernel by kernel, inside the default two-part solve (paired)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
echo "== paired: default | fused FFN without s_setprio (v4)"
timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/v4.so" 12 3 2>&1 | tail -4
for v in v4_ph v4_ows v4_qws v4_all; do
  echo "== paired: v4 | $v"
  timeout 300 python tools/ab_engines.py "STABLETTS_HIP_LIB=$ROOT/tools/ab/v4.so" "STABLETTS_HIP_LIB=$ROOT/tools/ab/$v.so" 10 3 2>&1 | tail -4
done
echo "== paired ragged: default | v4"
AB_RAGGED=1 timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/v4.so" 8 3 2>&1 | tail -3
echo "== paired ST_SPLIT=1: default | v4"
ST_SPLIT=1 timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/v4.so" 8 3 2>&1 | tail -3
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_setprio.txt
