#!/bin/bash
# Round 5, session 6: cache policy of the full-line row stores (sc1 = shipped) and of the per-lane LDS-DMA loads, inside the solve
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
echo "== one launch sequence (ST_SPLIT=1), class times: st1 = nt, st2 = sc1 nt, st3 = sc0 sc1, st4 = sc0 sc1 nt, dmant = LDS-DMA nt, both = dmant + st2"
timeout 900 python tools/class_times_libs.py default tools/ab/st1.so tools/ab/st2.so tools/ab/st3.so tools/ab/st4.so tools/ab/dmant.so tools/ab/both.so 2>&1
for v in st1 st2 st3 dmant both; do
  echo "== paired default parts: default | $v"
  timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/$v.so" 8 3 2>&1 | tail -4
done
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_store_policy.txt
