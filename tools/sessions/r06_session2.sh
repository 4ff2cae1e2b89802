#!/bin/bash
# Round 6, session 2: training step after the small-launch work (bias partials out of wgrad_tn, one reduce launch per GEMM / per block,
# maxima published by the producers, one memset per backward, dropout tables once per forward): tests, timing, kernel trace.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r06_s2_pytest_training.log; cat $OUT/r06_s2_pytest_training.log
for i in 1 2 3; do timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print({k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; done | tee $OUT/r06_s2_train_bench.txt
bash tools/profile_train.sh r06s2 > $OUT/r06_s2_profile_train.log 2>&1; head -60 $OUT/prof_train_r06s2/train_kernel_stats.txt
