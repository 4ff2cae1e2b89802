#!/bin/bash
# Round-4 end state: headline bench line (with extras), ragged / config-3 lines, kernel stats + PMC passes, training profile.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 > $OUT/r04_bench_final.json; cut -c1-400 $OUT/r04_bench_final.json
timeout 300 python bench.py --ragged --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/r04_bench_ragged_final.json; cut -c1-200 $OUT/r04_bench_ragged_final.json
timeout 300 python bench.py --n-timesteps 50 --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/r04_bench_config3_final.json; cut -c1-200 $OUT/r04_bench_config3_final.json
bash tools/profile_round.sh r04f > $OUT/r04_profile_round.log 2>&1; tail -40 $OUT/r04_profile_round.log
bash tools/profile_train.sh r04f > $OUT/r04_profile_train.log 2>&1; tail -5 $OUT/r04_profile_train.log
