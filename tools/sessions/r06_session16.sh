#!/bin/bash
# Round 6, session 16: time-MLP backward moved to the start of the last backward part; side streams at the lowest stream priority
# (ST_TRAIN_SIDE=2): paired against the library of the commit before.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
for i in 1 2 3 4; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06d.so tb before; tb tail_first; ST_TRAIN_SIDE=2 tb tail_first_low_prio_side; done 2>&1 | tee $OUT/r06_s16_tail_prio.txt
