"""GPU busy fraction and per-kernel totals of the steady state of a rocprofv3 kernel trace (dev tool).
usage: gap_stats.py <kernel_trace.csv> [tail fraction, default 0.5]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = []
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None:
        gaps.append(max(0, s - prev_end))
    prev_end = e if prev_end is None else max(prev_end, e)
print("kernels %d  wall %.3f ms  busy %.3f ms (%.1f%%)  mean gap %.2f us  gaps>10us: %d (%.3f ms)" % (
    len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(gaps) / len(gaps) / 1e3,
    sum(1 for g in gaps if g > 10000), sum(g for g in gaps if g > 10000) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[r["Kernel_Name"][:70]]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s %6d %10.1f us  avg %8.2f  %5.1f%%" % (k, n, t / 1e3, t / n / 1e3, 100.0 * t / busy))
# where the long gaps are: (kernel before -> kernel after), aggregated
loc = collections.defaultdict(lambda: [0, 0])
prev = None
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None and s - prev_end > 10000:
        a = loc[(prev["Kernel_Name"][:48], r["Kernel_Name"][:48])]
        a[0] += 1; a[1] += s - prev_end
    if prev_end is None or e > prev_end:
        prev_end = e
    prev = r
print("-- gaps > 10 us by location")
for (a, b), (n, t) in sorted(loc.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%5d x  avg %7.1f us   %s  ->  %s" % (n, t / n / 1e3, a, b))
