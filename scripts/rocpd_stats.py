#!/usr/bin/env python3
"""Kernel-stats summary (name, calls, total/avg/min/max ns, %) from a rocprofv3 rocpd sqlite database.
Equivalent of the `--stats` kernel table; usage: rocpd_stats.py results.db > profiles/<name>_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for n, c, t, a, mn, mx in rows:
    print(f"\"{n}\",{c},{t},{a:.1f},{mn},{mx},{100.0 * t / tot:.2f}")
