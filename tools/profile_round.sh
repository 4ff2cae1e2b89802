#!/bin/bash
# Round profile: rocprofv3 kernel stats + two PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench command.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
# per-kernel passes run the solve as ONE launch sequence (ST_SPLIT=1), like the loop bench.py samples the roofline kernel
# in: per-launch durations / bytes then describe whole-batch launches that do not share the chip with the other part
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_default -o kt -- $CMD > $OUT/kt_default.log 2>&1
export ST_SPLIT=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o s -- $CMD > $OUT/sq.log 2>&1
cd $ROOT
KS=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
FC=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
WC=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python tools/rocprof_summary.py stats $KS > $OUT/kernel_stats.txt
KD=$(find $OUT/kt_default -name '*kernel_stats.csv' | head -1)
python tools/rocprof_summary.py stats $KD > $OUT/kernel_stats_default_two_part.txt
python tools/rocprof_summary.py pmc $FC $WC $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
SC=$(find $OUT/sq -name '*counter_collection.csv' | head -1)
python tools/rocprof_summary.py mfma $SC $KS > $OUT/mfma_util.txt
ls $OUT; head -14 $OUT/mfma_util.txt; head -12 $OUT/kernel_stats.txt; head -12 $OUT/pmc_traffic.txt
