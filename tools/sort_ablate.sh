#!/bin/bash
# chunk-sort ablations (GSR_SORT_ABLATE builds in build_v_sa*): per-kernel events of the rasterizer on the bench scene
mkdir -p gpurun_out/r05sort
for v in "" 1 2 4 16 32 64; do
  if [ -z "$v" ]; then lib=""; else lib="lib_dir=build_v_sa$v"; fi
  echo "== GSR_SORT_ABLATE=${v:-0}"
  GA_DEV=$lib REPS=20 timeout 300 python tools/bench_raster.py 2>&1 | grep -v "^$" | sed -e 's/  */ /g'
done > gpurun_out/r05sort/ablate.txt 2>&1
cat gpurun_out/r05sort/ablate.txt
