#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite) kernel trace like `--stats` would:
    python tools/rocpd_stats.py results.db [top_n] > profiles/xxx_kernel_stats.txt"""
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
total = sum(r[2] for r in rows)
print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}")
for n, cnt, tot, avg, mn, mx in rows[:top]:
    print(f"{n[:90]:<90} {cnt:>7} {tot/1e6:>10.3f} {avg/1e3:>10.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {100*tot/total:>6.2f}")
print(f"{'TOTAL (all kernels)':<90} {sum(r[1] for r in rows):>7} {total/1e6:>10.3f}")
