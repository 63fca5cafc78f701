#!/usr/bin/env python
"""Find the step-sized window (after the first second of steady state) with the largest kernel time
that differs from a normal step and print its kernel histogram minus a normal window's (dev tool)."""
import sqlite3, sys, collections
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
# steady state: skip everything before the last long (>10 ms) kernel
last_long = max((s for n, s, e in rows if e - s > 10e6), default=t0)
rows = [(n, s, e) for n, s, e in rows if s > last_long + 200e6]
win = 50e6
buckets = collections.defaultdict(lambda: collections.Counter())
busy = collections.Counter()
for n, s, e in rows:
    b = int((s - t0) // win)
    buckets[b][n[:80]] += 1
    busy[b] += e - s
bs = sorted(buckets)
cnt = {b: sum(buckets[b].values()) for b in bs}
med = sorted(cnt.values())[len(cnt) // 2]
print("median kernels per 50 ms:", med)
for b in bs:
    if cnt[b] > 1.3 * med or busy[b] > 1.0 * win:
        print(f"window t={b*50} ms: {cnt[b]} kernels, busy {busy[b]/1e6:.1f} ms")
        norm = buckets[bs[len(bs)//2]]
        diff = {k: v - norm.get(k, 0) for k, v in buckets[b].items() if v - norm.get(k, 0) > 2}
        for k, v in sorted(diff.items(), key=lambda kv: -kv[1])[:15]:
            print(f"     +{v:5d}  {k}")
