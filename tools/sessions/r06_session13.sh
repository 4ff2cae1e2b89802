#!/bin/bash
# Round 6, session 13: attention backward with the per-query constants folded (5 instead of 7-8 vector instructions per element in dQ pass 2
# and dK/dV) + dQ pass 2's batched fragment reads: training tests (gates unchanged), paired step time against the library before.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{ timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu -s 2>&1 | grep -E "matched|worst|passed|failed|FAILED|B=64" | head -60
for i in 1 2 3 4; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06b.so tb before; tb aug_now; done
} 2>&1 | tee $OUT/r06_s13_attn_bwd_folded.txt
bash tools/profile_train.sh r06s13 > $OUT/r06_s13_profile_train.log 2>&1; grep -E "attn_bwd|attention_kernel" $OUT/prof_train_r06s13/train_kernel_stats.txt | cut -c1-120 | tee -a $OUT/r06_s13_attn_bwd_folded.txt
