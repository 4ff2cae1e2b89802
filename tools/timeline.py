#!/usr/bin/env python
"""Timeline analysis of a rocprofv3 --kernel-trace CSV of the bench command (developer tool): for the LAST solve in the trace, how
busy the GPU is (union of the kernels' intervals), how much of the time kernels of the two launch sequences (queues) overlap, the
idle gaps, and per kernel class the summed duration.     python tools/timeline.py gpurun_out/prof_r05/kt_default/kt_kernel_trace.csv"""
import csv, sys, collections

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
# a solve ends with from_time_major_kernel; take the window between the last two of them
ends = [i for i, r in enumerate(rows) if "from_time_major" in r[3]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1      # which solve of the trace (default: the last; bench.py's timed two-part solves come before its single-sequence sampling loop)
if which < 0: which += len(ends)
lo, hi = (ends[which - 1] + 1 if which > 0 else 0), ends[which] + 1
win = rows[lo:hi]
t0, t1 = win[0][0], max(r[1] for r in win)
span = (t1 - t0) / 1e3
# union of intervals, overlap depth
ev = []
for s, e, q, n in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0; depth = 0; last = t0; gaps = []
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    if depth == 0 and t - last > 0: gaps.append((t - last) / 1e3)
    depth += d; last = t
queues = collections.Counter(q for _, _, q, _ in win)
print(f"solve #{which} of {len(ends)} in the trace: {len(win)} kernels on queues {dict(queues)}, span {span:.1f} us")
print(f"GPU busy (>= 1 kernel running) {busy / 1e3:.1f} us = {busy / 1e3 / span:.3f}; >= 2 kernels running {over / 1e3:.1f} us = {over / 1e3 / span:.3f}")
print(f"idle gaps: {len(gaps)} totalling {sum(gaps):.1f} us; > 2 us: {sum(1 for g in gaps if g > 2)} totalling {sum(g for g in gaps if g > 2):.1f} us; longest {sorted(gaps)[-5:]}")
dur = collections.defaultdict(float); cnt = collections.Counter()
for s, e, q, n in win:
    k = n.split("(")[0].replace("void st::", "")[:48]
    dur[k] += (e - s) / 1e3; cnt[k] += 1
tot = sum(dur.values())
print(f"sum of kernel durations {tot:.1f} us = {tot / span:.3f} x the span")
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v:9.1f} us  {cnt[k]:4d} x {v / cnt[k]:7.1f}  {k}")
# what runs beside the fused FFN: for every FFN kernel, the time another kernel overlaps it, by class
beside = collections.defaultdict(float); ffn_t = 0.0
ffn = [(s, e) for s, e, q, n in win if "ffn_fused" in n or "ffn_wino" in n]
for fs, fe in ffn:
    ffn_t += fe - fs
    for s, e, q, n in win:
        if e <= fs or s >= fe or (s == fs and e == fe): continue
        k = n.split("(")[0].replace("void st::", "")[:32]
        beside[k] += min(e, fe) - max(s, fs)
print(f"while a fused-FFN kernel runs ({ffn_t / 1e3:.0f} us of kernel time), other kernels run beside it for:")
for k, v in sorted(beside.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {v / 1e3:9.1f} us  {k}")
