#!/bin/bash
# the reference's own loop: throughput now, and where the GPU idles in it (kernel trace -> tools/gap_stats.py)
mkdir -p gpurun_out/r05loop
python tools/reference_loop.py --iters 320 --out gpurun_out/r05loop/reference_loop.json 2>&1 | tail -1 | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/rl -o rl -- python $GRAFT_REPO_ROOT/tools/reference_loop.py --iters 200 --frames 32 > /tmp/rl.log 2>&1
f=$(find /tmp/rl -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/gap_stats.py $f 0.3 > gpurun_out/r05loop/gap_stats.txt 2>&1
head -3 gpurun_out/r05loop/gap_stats.txt; sed -n '/gaps > 10/,$p' gpurun_out/r05loop/gap_stats.txt
