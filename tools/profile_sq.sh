#!/bin/bash
# SQ wave-state counters per kernel (where the wave cycles of a kernel go): usage bash tools/profile_sq.sh <tag>
TAG=${1:-sq}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ST_SPLIT=1
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print(tag, "no output; log tail:"); print(open("$OUT/%s.log" % tag).read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:7]:
        wc = v.get("SQ_WAVE_CYCLES", 1.0)
        print(k, {c: round(x / wc, 3) if c != "SQ_WAVE_CYCLES" else x for c, x in v.items()})
PY
