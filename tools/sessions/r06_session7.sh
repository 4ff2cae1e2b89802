#!/bin/bash
# Round 6, session 7: batched per-item linears + two-launch attention operand copies, paired against the library of the commit before
# (tools/ab/lib_r06a.so); steady-state launch count from the rocprofv3 kernel trace of steps 4..8.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
for i in 1 2 3 4; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06a.so tb r06a_before; tb r06_now; done | tee $OUT/r06_s7_train_ab.txt
