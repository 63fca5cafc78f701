#!/bin/bash
# Evidence for every workload BASELINE.json names that one GPU can run, from ONE script:
# per workload the bench JSON line (no tracer) and the rocprofv3 --kernel-trace --stats table of the same command.
#   gpurun_out/<tag>/<name>.json            bench.py line
#   gpurun_out/<tag>/<name>_kernel_stats.txt per-kernel table (calls, total, average, share), kernel time per iteration
# usage: bash tools/collect_r04.sh [tag] [names...]      names: config2 stage2 config5 headline (default: all four)
R=$PWD; tag=${1:-r04}; shift; O=$R/gpurun_out/$tag; mkdir -p $O
names=${@:-config2 stage2 config5 headline}
declare -A ARGS=( [config2]="--config 2" [stage2]="--stage 2" [config5]="--config 5 --global-batch 1" [headline]="" )
# stage 2 settles later than stage 1 in a fresh process (bench.py: secondary_stage2_line): warm-up of 60 iterations
declare -A WARM=( [config2]=5 [stage2]=60 [config5]=60 [headline]=5 )
for n in $names; do
  a=${ARGS[$n]}
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup ${WARM[$n]} $a --no-fixed-batch > $O/$n.json 2> $O/$n.err
  python - $O/$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), d["config"]["workload"][:60])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_$n
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python $R/bench.py --steps 25 --warmup 10 $a --no-cpu-baseline --no-kernel-events --no-fixed-batch > /tmp/prof_$n.log 2>&1
    f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
    python - "$f" "$a" > $O/${n}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("rocprofv3 --kernel-trace --stats -- python bench.py --steps 25 --warmup 10 %s --no-cpu-baseline --no-kernel-events --no-fixed-batch  (35 iterations incl. warm-up)" % sys.argv[2])
print("%-100s %7s %11s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:70]:
    print("%-100s %7s %11.1f %9.2f %6.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time per iteration: %.1f us" % (tot / 1e3 / 35))
PY
    tail -1 $O/${n}_kernel_stats.txt )
done
