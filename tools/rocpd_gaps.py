#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd SQLite): how much of a tick is launch gaps.
usage: python tools/rocpd_gaps.py x_results.db [skip_first_n_kernels]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[skip:]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
# union of busy intervals (kernels on side streams overlap)
iv = sorted((s, e) for _, s, e in rows)
union, cs, ce = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs
        gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print(f"kernels {len(rows)}  span {span/1e6:.3f} ms  sum of durations {busy/1e6:.3f} ms  union busy {union/1e6:.3f} ms  idle {100*(span-union)/span:.2f} %")
if gaps:
    gaps.sort()
    print(f"gaps: n {len(gaps)}  mean {sum(gaps)/len(gaps)/1e3:.2f} us  p50 {gaps[len(gaps)//2]/1e3:.2f} us  p90 {gaps[int(len(gaps)*0.9)]/1e3:.2f} us  max {gaps[-1]/1e3:.2f} us")
ticks = sum(1 for n, _, _ in rows if "af_tick_kernel" in n)
if ticks:
    print(f"ticks {ticks}: span per tick {span/ticks/1e3:.1f} us, idle per tick {(span-union)/ticks/1e3:.1f} us")
