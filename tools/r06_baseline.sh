#!/bin/bash
# round 6: rasterizer micro-benchmark + size sweep (run through gpurun)
mkdir -p gpurun_out
python tools/bench_raster.py --out gpurun_out/${TAG:-r06a}_raster_ubench.json --sets ${SETS:-avatar_3mm,avatar_10mm,avatar_20mm,general,warmup_150,warmup_300} --sizes ${SIZES:-200k,300k} --seeds ${SEEDS:-2} --iters ${ITERS_N:-30} > gpurun_out/${TAG:-r06a}_raster_ubench.txt 2>&1
grep -v "^$\|Warning\|warn" gpurun_out/${TAG:-r06a}_raster_ubench.txt | tail -16
python - <<'PY'
import json,os
p="gpurun_out/%s_raster_ubench.json" % os.environ.get("TAG","r06a")
if os.path.exists(p):
    for r in json.load(open(p))["rows"]:
        print(r["size"], r["set"], r["frame0"])
PY
if [ -z "$NO_DSWEEP" ]; then
ITERS="${DS_ITERS:-7 150 300}" tools/dsweep.sh > gpurun_out/${TAG:-r06a}_dsweep.txt 2>&1
cat gpurun_out/${TAG:-r06a}_dsweep.txt
fi
