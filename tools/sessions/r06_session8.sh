#!/bin/bash
# Round 6, session 8: what the hi + lo V' operand pair of the attention backward costs and buys (ablation build -DST_ABL_NO_VLO):
# step time paired, and the gradient tests' matched-operand numbers with it.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{ for i in 1 2 3; do tb default; STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_novlo.so tb no_vlo; done
echo "== gradient tests with the ablation library (matched-operand lines)"
STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_novlo.so timeout 900 python -m pytest tests/test_gpu_training.py -q -m gpu -s -k "matched or config5 or trained_like or fixture" 2>&1 | grep -E "matched|worst|passed|failed|FAILED|Error" | head -60
echo "== the same lines with the shipped library"
timeout 900 python -m pytest tests/test_gpu_training.py -q -m gpu -s -k "matched or config5 or trained_like or fixture" 2>&1 | grep -E "matched|worst|passed|failed|FAILED" | head -60
} 2>&1 | tee $OUT/r06_s8_no_vlo.txt
