#!/usr/bin/env python
"""One steady-state training iteration as an ordered kernel list (dev tool): start offset, duration, gap to the previous
kernel's end, name. Iterations are delimited by adam_kernel. usage: iter_timeline.py <kernel_trace.csv> [iteration index from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = ends[-back - 1] + 1, ends[-back] + 1
it = rows[a:b]
t0 = int(it[0]["Start_Timestamp"]); prev_end = t0
small = 0.0; n_small = 0
for r in it:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    if d < 12: small += d; n_small += 1
    print("%8.1f  %7.2f  gap %6.2f  %s" % ((s - t0) / 1e3, d, (s - prev_end) / 1e3, r["Kernel_Name"][:110]))
    prev_end = max(prev_end, e)
print("kernels %d  span %.1f us  busy %.1f us | kernels under 12 us: %d, %.1f us" % (
    len(it), (prev_end - t0) / 1e3, sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in it) / 1e3, n_small, small))
