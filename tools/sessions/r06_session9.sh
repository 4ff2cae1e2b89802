#!/bin/bash
# Round 6, session 9: the attention backward compiled with -fno-slp-vectorize (no packed fp32 VALU beside the MFMAs), paired against the
# library of the commit before (tools/ab/lib_slp.so); training tests (gradients must be unchanged within the gates; the dropout test's
# bitwise checks included); per-kernel times.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{ for i in 1 2 3 4; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_slp.so tb slp_before; tb no_slp_now; done
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -4
} 2>&1 | tee $OUT/r06_s9_attn_bwd_noslp.txt
bash tools/profile_train.sh r06s9 > $OUT/r06_s9_profile_train.log 2>&1; grep -E "attn_bwd|attention_kernel" $OUT/prof_train_r06s9/train_kernel_stats.txt | cut -c1-120 | tee -a $OUT/r06_s9_attn_bwd_noslp.txt
