#!/bin/bash
# rocprofv3 --kernel-trace --stats of the stage-2 bench command -> per-kernel table; usage: tools/kstats_s2.sh [GA_DEV value]
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_ks2
GA_DEV=$1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks2 -o p -- python $R/bench.py --stage 2 --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events --no-fixed-batch --no-secondary --no-measure-traffic > /tmp/prof_ks2.log 2>&1
f=$(find /tmp/prof_ks2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
tot = 0.0
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: tot += float(r["TotalDurationNs"])
print("rocprofv3 --kernel-trace --stats -- python bench.py --stage 2 --steps 25 --warmup 10 ... (35 iterations incl. warm-up)")
print("%-84s %7s %11s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:70]:
    print("%-84s %7s %11.1f %9.2f %6.2f" % (r["Name"][:84], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time per iteration: %.1f us" % (tot / 1e3 / 35))
PY
