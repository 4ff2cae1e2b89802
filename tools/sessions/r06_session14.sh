#!/bin/bash
# Round 6, session 14: tiled weight-gradient plane reduction (coalesced 256-byte output runs) + vectorised q/k/v gradient pack: bitwise
# comparison with the library before, training tests, paired step time, per-kernel times.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_')})"; }
{ timeout 600 python tools/grad_bitwise_ab.py $ROOT/tools/ab/lib_r06c.so default 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3 4; do STABLETTS_HIP_LIB=$ROOT/tools/ab/lib_r06c.so tb before; tb now; done
} 2>&1 | tee $OUT/r06_s14_reduce_pack.txt
bash tools/profile_train.sh r06s14 > $OUT/r06_s14_profile_train.log 2>&1; grep -E "wgrad_reduce|qkv_grad_pack" $OUT/prof_train_r06s14/train_kernel_stats.txt | cut -c1-130 | tee -a $OUT/r06_s14_reduce_pack.txt
