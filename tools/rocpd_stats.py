#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace as a --stats style table.
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}  (rocprofv3 --kernel-trace --stats; durations in ns)")
print(f"{'Name':<100} {'Calls':>8} {'TotalNs':>14} {'AvgNs':>12} {'MinNs':>10} {'MaxNs':>10} {'Pct':>7}")
for n, c, t, a, mn, mx in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n[:100]:<100} {c:>8} {t:>14} {a:>12.0f} {mn:>10} {mx:>10} {100.0*t/tot:>6.2f}%")
print(f"{'TOTAL':<100} {sum(r[1] for r in rows):>8} {tot:>14}")
