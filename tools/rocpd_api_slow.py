#!/usr/bin/env python
"""Long host API calls from a rocprofv3 rocpd db recorded with --hip-trace (dev tool)."""
import sqlite3, sys
db, thr = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print("tables:", [t for t in tabs if "region" in t.lower() or "api" in t.lower() or "string" in t.lower()][:20])
for t in tabs:
    if t.lower().startswith("regions") or t.lower() == "regions":
        cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
        print(t, cols)
        break
try:
    rows = c.execute("select name, start, end from regions order by start").fetchall()
except Exception as e:
    print("regions query failed:", e)
    rows = []
if rows:
    t0 = rows[0][1]
    for n, s, e in rows:
        if (e - s) / 1e6 > thr:
            print(f"t={1e-6*(s-t0):10.1f} ms  dur {1e-6*(e-s):8.2f} ms  {str(n)[:100]}")
# which kernels were dispatched by the slow hipLaunchKernel calls (same thread, first dispatch after the call started)
try:
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("kernels cols:", kcols)
    name_col = "name" if "name" in kcols else "kernel_name"
    slow = c.execute("select id, name, start, end, tid, corr_id, stack_id from regions where (end-start) > ? order by start", (thr * 1e6,)).fetchall()
    for rid, n, s, e, tid, corr, stack in slow[-12:]:
        k = c.execute(f"select {name_col}, start from kernels where start >= ? order by start limit 1", (s,)).fetchone()
        k2 = None
        if "stack_id" in kcols:
            k2 = c.execute(f"select {name_col} from kernels where stack_id = ? limit 1", (stack,)).fetchone()
        print(f"t={1e-6*(s-t0):10.1f} {n} dur {1e-6*(e-s):7.1f} ms -> next kernel: {str(k[0])[:90] if k else None} | by stack: {str(k2[0])[:90] if k2 else None}")
except Exception as ex:
    print("join failed:", ex)
