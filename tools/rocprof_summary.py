"""Summarise rocprofv3 outputs (ROCm 7.2) as text for profiles/:
   python tools/rocprof_summary.py stats  <kernel_stats.csv | results.db>
   python tools/rocprof_summary.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv>
PMC: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced streams (MI355X_MICROARCH.md, HBM section) -> doubled below.  One counter per pass (TCC slot limits)."""
import collections
import csv
import json
import sqlite3
import sys


def stats(path):
    print(f"# rocprofv3 --kernel-trace --stats summary of {path} (durations in us)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    if path.endswith(".db"):
        rows = sqlite3.connect(path).execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    else:
        rows = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                 float(r["Percentage"])) for r in csv.DictReader(open(path))]
    for name, calls, tot, avg, pct in rows:
        print(f"{calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {name[:150]}")


def pmc(fetch_csv, write_csv):
    agg = collections.defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0, "wl": 0})
    for r in csv.DictReader(open(fetch_csv)):
        if r["Counter_Name"] == "FETCH_SIZE":
            a = agg[r["Kernel_Name"]]; a["launches"] += 1; a["fetch_kib"] += float(r["Counter_Value"])
    for r in csv.DictReader(open(write_csv)):
        if r["Counter_Name"] == "WRITE_SIZE":
            a = agg[r["Kernel_Name"]]; a["wl"] += 1; a["write_kib"] += float(r["Counter_Value"])
    out = {}
    print("# HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)")
    print(f"{'launches':>8} {'fetch_MB(x2 corrected)':>24} {'write_MB':>10} {'total_MB':>10}  kernel")
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["fetch_kib"] * 2 + kv[1]["write_kib"])):
        if not a["launches"] or not a["wl"]:
            continue
        f = 2.0 * a["fetch_kib"] / a["launches"] * 1024 / 1e6
        w = a["write_kib"] / a["wl"] * 1024 / 1e6
        out[k] = {"fetch_bytes_per_launch": f * 1e6, "write_bytes_per_launch": w * 1e6, "launches": a["launches"]}
        print(f"{a['launches']:8d} {f:24.1f} {w:10.1f} {f + w:10.1f}  {k[:120]}")
    return out


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        res = pmc(sys.argv[2], sys.argv[3])
        if len(sys.argv) > 4:
            json.dump(res, open(sys.argv[4], "w"), indent=1)
