#!/bin/bash
# Round-6 end state, ONE box: full GPU test suite + smoke; kernel stats + PMC passes first (r06_pmc_traffic.json carries the csrc digest and is
# put where bench.py reads it), then the headline bench line (every leg, both CPU thread counts), ragged / config-3 lines, the 2-rank line on the
# shared GPU; wave-state counters; training profile.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | tail -150 > $OUT/r06_pytest_gpu_final.log; tail -3 $OUT/r06_pytest_gpu_final.log
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 > $OUT/r06_smoke_final.log; cat $OUT/r06_smoke_final.log
bash tools/profile_round.sh r06 > $OUT/r06_profile_round.log 2>&1; tail -30 $OUT/r06_profile_round.log
cp $OUT/prof_r06/pmc_traffic.json $ROOT/profiles/r06_pmc_traffic.json; cp $OUT/prof_r06/pmc_traffic.txt $ROOT/profiles/r06_pmc_traffic.txt
timeout 900 python bench.py --steps 20 --warmup 3 --cpu-all-cores 2>$OUT/r06_bench_final.err | tail -1 > $OUT/r06_bench_final.json; cut -c1-300 $OUT/r06_bench_final.json
timeout 600 python bench.py --ragged --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/r06_bench_ragged_final.json; cut -c1-200 $OUT/r06_bench_ragged_final.json
timeout 300 python bench.py --n-timesteps 50 --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/r06_bench_config3_final.json; cut -c1-200 $OUT/r06_bench_config3_final.json
BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline 2>$OUT/r06_bench_2ranks.err | tail -1 > $OUT/r06_bench_2ranks_shared_gpu.json; cut -c1-200 $OUT/r06_bench_2ranks_shared_gpu.json
bash tools/profile_sq.sh r06sq > $OUT/r06_final_sq_counters.txt 2>&1; tail -16 $OUT/r06_final_sq_counters.txt
bash tools/profile_train.sh r06 > $OUT/r06_profile_train.log 2>&1; tail -5 $OUT/r06_profile_train.log
