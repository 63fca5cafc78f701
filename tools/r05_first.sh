#!/bin/bash
# round 5, first GPU call: the map-path kernels (tests), then A/B bench of decoder_map on / off
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decoder_map_gpu.py -q -x -m gpu > $O/test_map.log 2>&1; echo "map tests rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_decoder_map_gpu.py -q -m gpu > $O/test_map_all.log 2>&1; echo "map tests (all) rc=$?" >> $O/summary.txt
B="--steps 100 --warmup 20 --no-fixed-batch --no-secondary --no-cpu-baseline --no-measure-traffic"
timeout 600 python bench.py $B > $O/bench_map.json 2> $O/bench_map.err; echo "bench map rc=$?" >> $O/summary.txt
GA_DEV=decoder_map=0 timeout 600 python bench.py $B > $O/bench_nomap.json 2> $O/bench_nomap.err; echo "bench nomap rc=$?" >> $O/summary.txt
timeout 600 python bench.py $B --stage 2 > $O/bench_map_s2.json 2> $O/bench_map_s2.err; echo "bench s2 map rc=$?" >> $O/summary.txt
GA_DEV=decoder_map=0 timeout 600 python bench.py $B --stage 2 > $O/bench_nomap_s2.json 2> $O/bench_nomap_s2.err; echo "bench s2 nomap rc=$?" >> $O/summary.txt
timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py -q -x -m gpu > $O/test_fused_model.log 2>&1; echo "fused+model tests rc=$?" >> $O/summary.txt
python - <<'PY' >> $O/summary.txt
import json
for n in ("bench_map","bench_nomap","bench_map_s2","bench_nomap_s2"):
    try:
        d=json.loads(open(f"gpurun_out/r05a/{n}.json").read().strip().splitlines()[-1])
        k=d.get("kernels",{}).get("per_kernel",{})
        print(n, round(d["value"],1), "it/s", {a:round(b["us_per_iter"],1) for a,b in k.items()})
    except Exception as e:
        print(n, "ERR", e)
PY
cat $O/summary.txt; tail -30 $O/test_map.log
