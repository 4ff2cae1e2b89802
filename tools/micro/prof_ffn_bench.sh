#!/bin/bash
# rocprofv3 PMC passes over the fused-FFN micro-benchmark: matrix-pipe busy cycles, effective clock, wave-state counters per
# kernel variant.  usage on the GPU box: bash tools/micro/prof_ffn_bench.sh
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_ffnb
mkdir -p $OUT
BIN=$ROOT/tools/micro/ffn_bench
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BIN > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/b -o b -- $BIN > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -o a -- $BIN > $OUT/a.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections, re
def short(k):
    m = re.search(r"ffn_fused_kernel<st::OpF16, (\d+), (\d+)>", k)
    return "ABL %s VAR %s" % (m.group(1), m.group(2)) if m else k[:40]
dur = {}
fs = glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)
if fs:
    for r in csv.DictReader(open(fs[0])):
        dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
for tag in ("b", "a"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print(tag, "no output; log tail:"); print(open("$OUT/%s.log" % tag).read()[-800:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"): n[k] += 1
    for k, v in sorted(acc.items()):
        if "ABL" not in k: continue
        d = dur.get(k, (0, 0))[0]
        if tag == "b":
            ga = v["GRBM_GUI_ACTIVE"] / max(n[k], 1)
            print("%-16s %7.1f us  clock %.2f GHz  mfma busy %.3f  lds insts/launch %.0f  mfma insts/launch %.0f" % (
                k, d, ga / (d * 1e3) if d else 0, v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["GRBM_GUI_ACTIVE"] * 1024, 1), v["SQ_INSTS_LDS"] / max(n[k], 1), v["SQ_INSTS_MFMA"] / max(n[k], 1)))
        else:
            wc = v["SQ_WAVE_CYCLES"]
            print("%-16s" % k, {c.replace("SQ_", ""): round(x / wc, 3) for c, x in v.items() if c != "SQ_WAVE_CYCLES"})
PY
