#!/bin/bash
# Round-4 A/B of the fused FFN kernel (ffn_fused.h) inside one GPU session: parity test, per-class times of the headline solve
# with / without it (single launch sequence and default parts), ablation builds (tools/ab/abl*.so: ST_FFN_ABL bits 1 = no
# epilogue, 2 = no SiLU arithmetic), rocprofv3 kernel trace of the fused run.   usage: bash tools/ab_fused_ffn.sh [rounds]
R=${1:-2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "fused_ffn or phased_k_loop" 2>&1 | tail -15 > $OUT/r04_fused_test.log
cat $OUT/r04_fused_test.log
{
for i in $(seq $R); do
  for v in "ST_FUSED_FFN=0 ST_SPLIT=1" "ST_FUSED_FFN=1 ST_SPLIT=1" "ST_FUSED_FFN=0" "ST_FUSED_FFN=1"; do
    echo -n "[$v] "; env $v timeout 300 python tools/class_times.py 2>&1 | tail -1
  done
done
for lib in abl1 abl2 abl3; do
  [ -f tools/ab/$lib.so ] && { echo -n "[$lib ST_SPLIT=1] "; STABLETTS_HIP_LIB=$ROOT/tools/ab/$lib.so ST_SPLIT=1 timeout 300 python tools/class_times.py 2>&1 | tail -1; }
done
} | tee $OUT/r04_ab_fused_ffn.txt
cd /tmp && export TMPDIR=/tmp
ST_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fused/kt -o kt -- python $ROOT/tools/class_times.py > $OUT/prof_fused_kt.log 2>&1
cd $ROOT
KS=$(find $OUT/prof_fused/kt -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && python tools/rocprof_summary.py stats $KS | head -30 | tee $OUT/r04_fused_kernel_stats.txt
