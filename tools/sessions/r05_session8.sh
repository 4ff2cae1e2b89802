#!/bin/bash
# Round 5, session 8: the fused FFN's loop variants (ST_FFN_VAR: 1 = LDS-DMA issue between the MFMA halves, 3 = + slab p+4 in flight, 4 = no s_setprio,
# 8 = software-pipelined phases without the group stagger) inside the default two-part solve, paired; solve-part count
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for v in 1 3 4 8; do
  echo "== paired default parts: default | ST_FFN_VAR=$v"
  timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/ffnvar$v.so" 8 3 2>&1 | tail -4
done
for i in 1 2; do for sp in 1 2 3 4; do echo -n "[ST_SPLIT=$sp] "; ST_SPLIT=$sp timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-40; done; done
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_ffn_var_parts.txt
