#!/usr/bin/env python
"""API calls and kernels inside the spike window printed by tools/step_times.py (dev tool).
usage: rocpd_window.py db 'SPIKE_WINDOW ...line...'"""
import sqlite3, sys, re, collections
db, line = sys.argv[1], sys.argv[2]
vals = dict(re.findall(r"(\w+)=(\d+)", line))
c = sqlite3.connect(db)
lo, hi = c.execute("select min(start), max(end) from regions").fetchone()
clock = None
for k in ("CLOCK_MONOTONIC", "CLOCK_MONOTONIC_RAW", "CLOCK_BOOTTIME", "CLOCK_REALTIME"):
    if lo <= int(vals[k]) <= hi + 5e9:
        clock = k
        break
print("trace range", lo, hi, "matching clock:", clock)
if clock is None:
    sys.exit()
end = int(vals[clock]); start = end - int(vals["dur_ns"]) - int(3e6)
rows = c.execute("select name, start, end from regions where start >= ? and start <= ? order by (end-start) desc limit 12", (start, end)).fetchall()
print("longest API calls in the window:")
for n, s, e in rows:
    print(f"   +{1e-6*(s-start):8.2f} ms  dur {1e-6*(e-s):8.3f} ms  {n}")
k = c.execute("select name, start, end from kernels where start >= ? and start <= ? order by start", (start, end)).fetchall()
print(len(k), "kernels in window; busy", sum(e - s for n, s, e in k) / 1e6, "ms")
cnt = collections.Counter(n[:70] for n, s, e in k)
for n, v in cnt.most_common(12):
    print(f"   {v:5d} {n}")
# biggest idle gaps
gaps = sorted(((k[i + 1][1] - k[i][2], k[i][0][:60], k[i + 1][0][:60]) for i in range(len(k) - 1)), reverse=True)[:5]
for g, a, b in gaps:
    print(f"   gap {g/1e6:8.2f} ms between {a} -> {b}")
