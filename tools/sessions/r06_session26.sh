#!/bin/bash
# Round 6, session 26: after templating the attention-backward operand copies on the hi + lo v mode: the default step against the library of the
# commit before (must not have moved), the mode's cost; the inference headline with both libraries (--dev-env marks the line).
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
hb() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 10 --dev-env 2>/dev/null | tail -1 | python -c "import json,sys; print('$1', round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
{
timeout 900 python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -2
for i in 1 2 3; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06f.so tb before; tb now_default; ST_TRAIN_VLO=1 tb now_v_hi_lo; done
for i in 1 2 3; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06f.so hb headline_before; hb headline_now; done
} 2>&1 | tee $OUT/r06_s26_v_hi_lo_optin.txt
