#!/bin/bash
# the round's GPU checks in one call: full -m gpu suite, smoke, bench (driver command) -> gpurun_out/<tag>/
tag=${1:-check}; O=gpurun_out/$tag; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; python - $O/bench_driver_cmd.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3),
      "fixed", d.get("fixed_global_batch", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"))
print({k: (round(v["us_per_iter"]), v["launches_per_iter"]) for k, v in d["kernels"]["per_kernel"].items()})
PY
tail -2 $O/bench.err
