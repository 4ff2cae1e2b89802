#!/bin/bash
# Round-4 (second session) GPU call 1: the fused FFN on 16x16x32 fragments (ffn_fused16.h) -- parity tests, the stand-alone
# micro-benchmark (32x32x16 and 16x16x32 kernels interleaved; random / model-like / zero data), per-class times of the headline
# solve with ST_FUSED_FFN=1 and 2 (single launch sequence and default parts), the bench line.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -s -k "fused_ffn" 2>&1 | tail -40 > $OUT/r04b_fused16_test.log
cat $OUT/r04b_fused16_test.log
{
for fill in 0 2 1; do timeout 120 tools/micro/ffn_bench 64 1000 $fill 1; done
} 2>&1 | tee $OUT/r04b_ffn_microbench_16x16x32.txt
{
for i in 1 2; do
  for v in "ST_FUSED_FFN=1 ST_SPLIT=1" "ST_FUSED_FFN=2 ST_SPLIT=1" "ST_FUSED_FFN=1" "ST_FUSED_FFN=2"; do
    echo -n "[$v] "; env $v timeout 300 python tools/class_times.py 2>&1 | tail -1
  done
done
} | tee $OUT/r04b_ab_fused16.txt
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee $OUT/r04b_bench_fused16.json
