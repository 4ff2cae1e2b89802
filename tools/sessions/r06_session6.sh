#!/bin/bash
# Round 6, session 6: batched per-item linears + fused attention operand copies: training tests, paired timing, launch count, and where the
# ATen fills / copies of a step come from.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/r06_s6_pytest_training.log
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
for i in 1 2 3; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r05.so tb r05; tb r06; done | tee $OUT/r06_s6_train_ab.txt
timeout 300 python tools/train_torch_ops.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_s6_torch_ops.txt | head -50
bash tools/profile_train.sh r06s6 > $OUT/r06_s6_profile_train.log 2>&1
python - <<'PY'
import re
tot=0;n=0
for l in open('gpurun_out/prof_train_r06s6/train_kernel_stats.txt'):
    m=re.match(r'\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)',l)
    if m: n+=int(m.group(1)); tot+=float(m.group(2))
print('launches per step',n/8,'kernel ms per step',tot/8e3)
PY
