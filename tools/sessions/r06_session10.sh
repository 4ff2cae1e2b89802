#!/bin/bash
# Round 6, session 10: dQ pass 2 of the attention backward with its fragment reads batched in front of the MFMAs: bitwise identity with the
# library before, paired step time, per-kernel times.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{ timeout 600 python tools/grad_bitwise_ab.py $ROOT/tools/ab/lib_slp.so default 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2 3 4; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_slp.so tb before; tb batched_now; done
} 2>&1 | tee $OUT/r06_s10_dq2_batched.txt
bash tools/profile_train.sh r06s10 > $OUT/r06_s10_profile_train.log 2>&1; grep -E "attn_bwd|attention_kernel" $OUT/prof_train_r06s10/train_kernel_stats.txt | cut -c1-120 | tee -a $OUT/r06_s10_dq2_batched.txt
