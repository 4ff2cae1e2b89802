#!/bin/bash
# Round 6, session 25: hi + lo v operands as an opt-in mode (ST_TRAIN_VLO=1): full GPU suite at the default, the mode's own test, step time
# at the default against the library of the commit before (nothing may have moved) and with the mode on.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_training.py -q -s -k "hi_lo_operand" 2>&1 | grep "f16, v\|passed\|failed"
for i in 1 2 3; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06f.so tb before; tb now_default; ST_TRAIN_VLO=1 tb now_v_hi_lo; done
for i in 1 2; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06f.so timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; print('before', round(json.loads(sys.stdin.read())['ms_per_step'],3))" ; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; print('now', round(json.loads(sys.stdin.read())['ms_per_step'],3))"; done
} 2>&1 | tee $OUT/r06_s25_v_hi_lo_optin.txt
