"""Summarise rocprofv3 outputs (ROCm 7.2) as text for profiles/:
   python tools/rocprof_summary.py stats  <kernel_stats.csv | results.db>
   python tools/rocprof_summary.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv>
   python tools/rocprof_summary.py mfma   <sq_counter_collection.csv> <kernel_stats.csv>
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


def _csrc_digest():
    """Digest of the kernel sources + build flags the profiled library was built from (stabletts_amd.build): bench.py prints it next
    to the digest of the sources it runs, so a table taken from older kernels is visible in the bench line."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from stabletts_amd import build
        return build._digest()[:16]
    except Exception:
        return None


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
    # whole-solve HBM traffic: every kernel of the run, divided by the number of solves (one from_time_major per solve)
    solves = max([v["launches"] for k, v in out.items() if "from_time_major" in k] or [1])
    total = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in out.values())
    out["_summary"] = {"solves": solves, "hbm_bytes_per_solve": total / solves, "csrc_digest": _csrc_digest()}
    print(f"# {solves} solves in the run, {total / solves / 1e9:.2f} GB of HBM traffic per solve (all kernels)")
    return out


def mfma(pmc_csv, stats_csv):
    """Matrix-pipe occupancy per kernel: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs of the chip;
    32 per v_mfma_f32_32x32x16) / (GRBM_GUI_ACTIVE x 1024 SIMDs).  GRBM_GUI_ACTIVE is reported per XCD and summed by
    rocprofv3 over the 8 XCDs on this stack (detected from the implied clock: active / duration must be < 3 GHz)."""
    dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(stats_csv))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(pmc_csv)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
    print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES (per launch, means)")
    print(f"{'launches':>8} {'mfma_busy_Mcyc':>15} {'gui_active_kcyc':>16} {'clock_GHz':>10} {'mfma_util':>10} {'vs_2.4GHz_peak':>15}  kernel")
    rows = []
    for k, a in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in a or "GRBM_GUI_ACTIVE" not in a or k not in dur:
            continue
        n = cnt[k]["GRBM_GUI_ACTIVE"]
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
        act = a["GRBM_GUI_ACTIVE"] / n
        clock = act / dur[k]                      # cycles per ns = GHz
        if clock > 3.0:                           # summed over the 8 XCDs
            act /= 8.0; clock /= 8.0
        if busy <= 0:
            continue
        util = busy / (act * 1024.0)
        peak = busy / (dur[k] * 2.4 * 1024.0)     # against the 2.4 GHz nominal clock the 2.5 PFLOP/s peak assumes
        rows.append((busy, n, act, clock, util, peak, k))
    for busy, n, act, clock, util, peak, k in sorted(rows, reverse=True):
        print(f"{n:8d} {busy / 1e6:15.2f} {act / 1e3:16.1f} {clock:10.2f} {util:10.3f} {peak:15.3f}  {k[:110]}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3])
    else:
        res = pmc(sys.argv[2], sys.argv[3])
        if len(sys.argv) > 4:
            json.dump(res, open(sys.argv[4], "w"), indent=1)
