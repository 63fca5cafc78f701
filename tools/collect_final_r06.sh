#!/bin/bash
# Round 6's judged artefacts in one GPU call -> gpurun_out/<tag>/ (copy what is kept into profiles/r06_*):
#   everything tools/collect_final_r04.sh collects (full -m gpu suite, smoke, the driver's bench command, kernel stats + gaps,
#   PMC traffic, config 2 / stage 2 / config 5 lines + kernel tables, the reference's train.py verbatim, a 2000-step run)
#   + the rasterizer micro-benchmark of SURVEY section 8d with PMC traffic (tools/bench_raster.py)
#   + the rasterizer against Gaussian size inside the training iteration (tools/dsweep.sh)
# usage: bash tools/collect_final_r06.sh r06_final
tag=${1:-r06_final}
bash tools/collect_final_r04.sh $tag
O=gpurun_out/$tag; mkdir -p $O
python tools/bench_raster.py --pmc --out $O/raster_ubench.json > $O/raster_ubench.txt 2>&1; grep "^[23]00k" $O/raster_ubench.txt
ITERS="7 60 150 300" tools/dsweep.sh > $O/dsweep.txt 2>&1; cat $O/dsweep.txt
python tools/cpu_enqueue.py > $O/cpu_enqueue.txt 2>&1; SYNC=1 python tools/cpu_enqueue.py > $O/cpu_enqueue_sync.txt 2>&1; tail -3 $O/cpu_enqueue_sync.txt
tools/ubench_atomic.sh $O 2>/dev/null
