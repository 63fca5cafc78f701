#!/usr/bin/env python
"""List kernels longer than a threshold (ms) and idle gaps between consecutive kernels from a rocpd db."""
import sqlite3, sys
db, thr = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
prev_end = rows[0][2]
for n, s, e in rows:
    if (e - s) / 1e6 > thr:
        print(f"t={1e-6*(s-t0):10.1f} ms  dur {1e-6*(e-s):8.2f} ms  {n[:100]}")
    if (s - prev_end) / 1e6 > thr:
        print(f"t={1e-6*(prev_end-t0):10.1f} ms  GAP {1e-6*(s-prev_end):8.2f} ms before {n[:80]}")
    prev_end = max(prev_end, e)
