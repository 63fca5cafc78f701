#!/usr/bin/env python
"""Average PMC counter values per kernel from a rocprofv3 counter_collection CSV.
usage: pmc_summary.py <csv> [substring filters...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2:] or [""]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if any(f in k for f in filt):
        agg[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        tail = v[len(v) // 2:]          # skip warm-up launches
        print(f"    {c:28s} mean {sum(tail)/len(tail):16.1f}   launches {len(v)}")
