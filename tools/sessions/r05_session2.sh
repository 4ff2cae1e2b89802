#!/bin/bash
# Round 5, session 2: full GPU suite + smoke on the new default (direct fused FFN), bench lines (headline, ragged)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v Warning > $OUT/r05_pytest_gpu_s2.log; tail -5 $OUT/r05_pytest_gpu_s2.log
grep -E "trained-like|max \|u\||ada_std" $OUT/r05_pytest_gpu_s2.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep smoke | tee $OUT/r05_smoke_s2.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-extras 2>&1 | tail -1 > $OUT/r05_bench_s2.json; cut -c1-300 $OUT/r05_bench_s2.json
python -c "import json; d=json.load(open('$OUT/r05_bench_s2.json')); print(d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline'])"
timeout 300 python bench.py --ragged --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 > $OUT/r05_bench_ragged_s2.json; cut -c1-200 $OUT/r05_bench_ragged_s2.json
