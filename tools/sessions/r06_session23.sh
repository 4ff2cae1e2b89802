#!/bin/bash
# Round 6, session 23: which kernels carry the 0.8 ms dropout costs the training step -- kernel stats with and without dropout, ST_TRAIN_SIDE=0
# (durations not inflated by the side stream's kernels sharing the chip).
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
ST_TRAIN_SIDE=0 bash tools/profile_train.sh r06drop > /dev/null 2>&1
ST_TRAIN_SIDE=0 bash tools/profile_train.sh r06nodrop --no-dropout > /dev/null 2>&1
python - <<'PY' | tee $OUT/r06_s23_dropout_by_kernel.txt
import re
def load(p):
    d = {}
    for ln in open(p):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", ln)
        if m: d[m.group(5).strip()[:90]] = (int(m.group(1)), float(m.group(2)))
    return d
a = load("gpurun_out/prof_train_r06drop/train_kernel_stats.txt"); b = load("gpurun_out/prof_train_r06nodrop/train_kernel_stats.txt")
print("per step (8 profiled steps), ST_TRAIN_SIDE=0: kernel, ms with dropout, ms without, difference")
rows = []
for k in sorted(set(a) | set(b)):
    ta = a.get(k, (0, 0.0))[1] / 8e3; tb = b.get(k, (0, 0.0))[1] / 8e3
    if abs(ta - tb) > 0.01: rows.append((ta - tb, k, ta, tb))
for d, k, ta, tb in sorted(rows, reverse=True): print(f"{ta:8.3f} {tb:8.3f} {d:+8.3f}  {k}")
print("total", sum(v[1] for v in a.values()) / 8e3, sum(v[1] for v in b.values()) / 8e3)
PY
