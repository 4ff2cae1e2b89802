#!/bin/bash
# Round 5, session 1: the shipped default on trial with trained-like weights at benchmark size; the recorded value of the
# re-reference test; per-class ragged / all-ones table; a bench line of this box.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python tools/parity_trained.py 1 3 2>&1 | grep -v Warning | tee $OUT/r05_parity_trained.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "re_reference or strong_gates" 2>&1 | grep -E "ada_std|passed|failed|Error" | tee $OUT/r05_re_reference_values.txt
{
for i in 1 2; do
  for r in 0 1; do
    echo -n "[ragged=$r] "; env CLASS_TIMES_RAGGED=$r timeout 200 python tools/class_times.py 2>&1 | tail -1
    echo -n "[ragged=$r ST_SPLIT=1] "; env CLASS_TIMES_RAGGED=$r ST_SPLIT=1 timeout 200 python tools/class_times.py 2>&1 | tail -1
  done
done
} | tee $OUT/r05_class_times_ragged.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-extras 2>&1 | tail -1 > $OUT/r05_bench_s1.json; cut -c1-600 $OUT/r05_bench_s1.json
