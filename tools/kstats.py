#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 rocpd database (developer tool): python tools/kstats.py <results.db> [runs]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
runs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {runs:g} runs = {tot/1e6/runs:.3f} ms/run")
for r in rows[:30]:
    print(f"{r[0][:96]:96s} {r[1]/runs:7.1f}/run {r[2]/1e3:8.2f} us {100*r[3]/tot:5.1f} %")
