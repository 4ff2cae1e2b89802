#!/bin/bash
# Round 6, session 27: ST_TRAIN_VLO=2 -- v as a hi + lo pair computed from h1 as a hi + lo pair (v = W_v h_hi + W_v h_lo): q/k gradients end to end,
# every gradient test under the mode, cost.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{
timeout 600 python -m pytest tests/test_gpu_training.py -q -s -k "hi_lo_operand" 2>&1 | grep "f16, \|passed\|failed\|Error" | cut -c1-250
echo "== the gradient tests under ST_TRAIN_VLO=2"
ST_TRAIN_VLO=2 timeout 1500 python -m pytest tests/test_gpu_training.py -q -s -k "gradients or trajectory_at_config5 or dropout or bitwise_neutral" 2>&1 | grep -v amdgpu.ids | grep "worst\|cosine\|passed\|failed\|T=1000" | cut -c1-250
for i in 1 2 3; do tb default; ST_TRAIN_VLO=1 tb v_hi_lo; ST_TRAIN_VLO=2 tb v_and_h1_hi_lo; done
} 2>&1 | tee $OUT/r06_s27_vlo2.txt
