#!/bin/bash
# Round 6, session 28: the q / k blocks of the q/k/v GEMM stop at c0 (their second-source weights are zero): training tests, step cost of the default
# (v from h1 as a hi + lo pair) against ST_TRAIN_VLO=0.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{
timeout 900 python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -2
for i in 1 2 3; do ST_TRAIN_VLO=0 tb single_operands; tb default_v_from_h1_pairs; done
} 2>&1 | tee $OUT/r06_s28_vlo2_kbound.txt
