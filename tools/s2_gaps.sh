#!/bin/bash
# stage 2: kernel-time sum per iteration and where the GPU idles (kernel trace -> tools/gap_stats.py)
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_s2g
GA_DEV=$1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2g -o p -- python $R/bench.py --stage 2 --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events --no-fixed-batch --no-secondary --no-measure-traffic > /tmp/prof_s2g.log 2>&1
f=$(find /tmp/prof_s2g -name "*kernel_trace.csv" | head -1)
mkdir -p $R/gpurun_out/r05s2g
python $R/tools/gap_stats.py $f 0.5 > $R/gpurun_out/r05s2g/gap_stats.txt
head -1 $R/gpurun_out/r05s2g/gap_stats.txt; sed -n '/gaps > 10/,$p' $R/gpurun_out/r05s2g/gap_stats.txt | head -24
tail -1 /tmp/prof_s2g.log | cut -c1-200
