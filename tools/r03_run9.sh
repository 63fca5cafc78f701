#!/bin/bash
timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_raster_hardening_gpu.py -q 2>&1 | grep -E "^E  .*Assert|FAILED|passed|failed" | cut -c1-250 | head
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o p -- python $R/bench.py --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events > /tmp/prof_r.log 2>&1
f=$(find /tmp/prof_r -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("render","seg_","preprocess","tile_","scatter","clear_frames","gather")): print("%-60s %6s %10.2f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
