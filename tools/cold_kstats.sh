#!/bin/bash
# rocprofv3 kernel table of the FIRST GPU process on a fresh box (cold vendor-library caches): usage tools/cold_kstats.sh <out.txt> <bench args...>
out=$1; shift
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_cold
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cold -o p -- python $R/bench.py "$@" --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events > /tmp/cold.log 2>&1
f=$(find /tmp/prof_cold -name "*kernel_stats.csv" | head -1)
python - $f "$*" > $R/$out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("first GPU process on a fresh box: rocprofv3 --kernel-trace --stats -- python bench.py %s --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events (35 iterations)" % sys.argv[2])
for r in rows[:40]:
    print("%-100s %7s %11.1f %9.2f %6.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time per iteration: %.1f us" % (tot / 1e3 / 35))
PY
