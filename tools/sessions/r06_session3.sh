#!/bin/bash
# Round 6, session 3: side streams of the backward -- bitwise test, the training tests, paired timing against the round-5 library
# (tools/ab/lib_r05.so, built from commit 60cb8fc) and against the single-stream order, alternating in one call.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r06_s3_pytest_training.log; cat $OUT/r06_s3_pytest_training.log
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
for i in 1 2 3; do
  STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r05.so tb r05
  ST_TRAIN_SIDE=0 tb r06_single_stream
  tb r06_side_streams
done | tee $OUT/r06_s3_train_ab.txt
bash tools/profile_train.sh r06s3 > $OUT/r06_s3_profile_train.log 2>&1; head -40 $OUT/prof_train_r06s3/train_kernel_stats.txt
