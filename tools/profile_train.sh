#!/bin/bash
# rocprofv3 kernel stats of the config-5 training step (tools/train_bench.py).  usage: bash tools/profile_train.sh <tag> [train_bench.py options]
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/tools/train_bench.py --steps 3 "${@:2}" > $OUT/kt.log 2>&1
cd $ROOT
KS=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
python tools/rocprof_summary.py stats $KS > $OUT/train_kernel_stats.txt
head -45 $OUT/train_kernel_stats.txt; tail -3 $OUT/kt.log
