#!/usr/bin/env python
"""Per-dispatch durations (us) of the last N kernel dispatches in a rocprofv3 rocpd database, in start order.
usage: python tools/rocpd_layers.py x_results.db [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
extra = [c for c in ("grid_size_x", "grid_size", "workgroup_size_x", "lds_block_size") if c in cols]
rows = cur.execute(f"select {', '.join([name_col, 'start', 'end'] + extra)} from kernels order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
t0 = rows[-n][1]
for r in rows[-n:]:
    print(f"{(r[1]-t0)/1e3:9.1f} us  +{(r[2]-r[1])/1e3:8.1f} us  {r[0][:60]:<60} {dict(zip(extra, r[3:]))}")
