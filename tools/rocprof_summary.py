"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as text:
   python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path} (durations in us)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) < 150 else name[:147] + "..."
        print(f"{calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1])
