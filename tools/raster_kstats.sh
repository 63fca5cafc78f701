#!/bin/bash
# per-kernel table (rocprofv3 --kernel-trace --stats) of the rasterizer micro-benchmark's child run of ONE set
# usage: tools/raster_kstats.sh <set> [size]      e.g. tools/raster_kstats.sh avatar_20mm 200k
R=$PWD; S=${1:-avatar_3mm}; Z=${2:-200k}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_rk
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rk -o p -- python $R/tools/bench_raster.py --child $S $Z > /tmp/prof_rk.log 2>&1
f=$(find /tmp/prof_rk -name "*kernel_stats.csv" | head -1)
python - "$f" "$S" "$Z" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gsr::" in r["Name"] or "clear_frames" in r["Name"] or "batch_status" in r["Name"]]
print("set", sys.argv[2], sys.argv[3], "(12 forward + backward calls, 2 frames per launch)")
for r in rows:
    print("%-60s %5s calls  avg %8.1f us" % (r["Name"].replace("gsr::(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
