#!/bin/bash
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_decoder_map_gpu.py -q -m gpu 2>&1 | tail -15 > $O/test_map.log; cat $O/test_map.log
timeout 300 python tools/microbench_upz.py 2>&1 | tee $O/microbench.log
B="--steps 100 --warmup 20 --no-fixed-batch --no-secondary --no-cpu-baseline --no-measure-traffic"
timeout 600 python bench.py $B > $O/bench_map.json 2> $O/bench_map.err
GA_DEV=decoder_map=0 timeout 600 python bench.py $B > $O/bench_nomap.json 2> $O/bench_nomap.err
python - <<'PY'
import json
for n in ("bench_map","bench_nomap"):
    try:
        d=json.loads(open(f"gpurun_out/r05c/{n}.json").read().strip().splitlines()[-1])
        k=d.get("kernels",{}).get("per_kernel",{})
        print(n, round(d["value"],1), "it/s", {a:(round(b["launches_per_iter"]),round(b["avg_us"],1)) for a,b in k.items() if a in ("mlp_fwd","layer_bwd","rowgemm","upsample_z_fwd","dz_upsample_t","wgrad_act")})
    except Exception as e:
        print(n, "ERR", e)
PY
