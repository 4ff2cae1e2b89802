#!/bin/bash
# Round 6, session 24: v as a hi + lo operand pair in the training forward (ST_TRAIN_VLO, default on): gradient parity lines with and without,
# step time with and without.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{
echo "== hi + lo v (default)"
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -s 2>&1 | grep -v amdgpu.ids | grep "worst\|cosine\|passed\|failed\|Error\|error\|matched\|trained-like\|T=1000" | cut -c1-260
echo "== ST_TRAIN_VLO=0"
ST_TRAIN_VLO=0 timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -s -k "gradients" 2>&1 | grep -v amdgpu.ids | grep "worst\|cosine\|passed\|failed\|matched\|trained-like" | cut -c1-260
for i in 1 2 3; do tb v_hi_lo; ST_TRAIN_VLO=0 tb v_one_operand; done
} 2>&1 | tee $OUT/r06_s24_v_hi_lo.txt
