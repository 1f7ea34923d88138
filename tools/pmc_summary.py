#!/usr/bin/env python
"""Per-kernel mean of rocprofv3 --pmc counters from a rocpd SQLite db. usage: pmc_summary.py x.db [name-filter]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # only the last N dispatches of each kernel (steady state)
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("# columns:", cols, file=sys.stderr)
name_col = "kernel_name" if "kernel_name" in cols else "name"
rows = cur.execute(f"select {name_col}, counter_name, value, dispatch_id from counters_collection order by dispatch_id").fetchall()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v, d in rows:
    if flt in k: acc[k][c].append(v)
for k in acc:
    print(k[:200])
    for c, vs in sorted(acc[k].items()):
        total = len(vs)
        vs = vs[-last:] if last else vs
        print(f"   {c:<34} n={len(vs):<5} mean={sum(vs)/len(vs):.4g} dispatches={total}")
