#!/bin/bash
# sustained vs burst: a long bench run with a per-100-step rate series, GPU clocks / power sampled beside it
# usage: bash tools/sustained.sh <outdir> [steps]
O=${1:-gpurun_out/sustained}; S=${2:-2000}; mkdir -p $O
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|mclk|Power|Temperature \(Sensor edge|junction' | tr -s ' ' | tr '\n' '|')"; sleep 0.5; done ) > $O/smi.txt 2>/dev/null &
SMI=$!
python bench.py --steps $S --warmup 30 --series 100 --no-cpu-baseline > $O/bench_series.json 2> $O/bench_series.err
kill $SMI
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_series.json"))
print("value", round(d["value"], 1), "it/s over", d["steps"], "steps;", "series:", d["config"]["series"]["iters_per_s"])
PY
grep -c . $O/smi.txt; head -3 $O/smi.txt; tail -2 $O/smi.txt
