#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command -> per-kernel table (stdout); usage: tools/kstats.sh [filter words...]
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o p -- python $R/bench.py --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events --no-fixed-batch --no-secondary > /tmp/prof_ks.log 2>&1
f=$(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1)
python - "$f" "$@" <<'PY'
import csv, sys
flt = sys.argv[2:]
tot = 0.0
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: tot += float(r["TotalDurationNs"])
print("rocprofv3 --kernel-trace --stats -- python bench.py --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events --no-fixed-batch --no-secondary  (35 iterations incl. warm-up)")
print("%-84s %7s %11s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:60]:
    n = r["Name"]
    if flt and not any(k in n for k in flt): continue
    print("%-84s %7s %11.1f %9.2f %6.2f" % (n[:84], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time per iteration: %.1f us" % (tot / 1e3 / 35))
PY
