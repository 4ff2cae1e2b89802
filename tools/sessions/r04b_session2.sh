#!/bin/bash
# GPU call 2: which fused-FFN kernel does the engine launch under ST_FUSED_FFN=1 / 2 (rocprofv3 kernel trace of one class survey),
# and the micro-benchmark with both weight-stream layouts packed from the same fp32 weights (outputs compared) in both run orders.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
{ timeout 120 tools/micro/ffn_bench 64 1000 0 2; timeout 120 tools/micro/ffn_bench 64 1000 2 2; } 2>&1 | tee $OUT/r04b_ffn_microbench_16x16x32_v2.txt
cd /tmp && export TMPDIR=/tmp
for f in 2 1; do
  ST_FUSED_FFN=$f ST_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f$f/kt -o kt -- python $ROOT/tools/class_times.py > $OUT/prof_f$f.log 2>&1
  tail -1 $OUT/prof_f$f.log
  KS=$(find $OUT/prof_f$f/kt -name '*kernel_stats.csv' | head -1)
  [ -n "$KS" ] && python $ROOT/tools/rocprof_summary.py stats $KS | head -12 | tee $OUT/r04b_kernel_stats_fused$f.txt
done
