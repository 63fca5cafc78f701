#!/bin/bash
O=gpurun_out/r03_5; mkdir -p $O
bash tools/sustained.sh $O 2000
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('20-step', d['value'])"
python tools/cpu_enqueue.py 2>&1 | grep -v amdgpu | tee $O/cpu_enqueue.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o p -- python $R/bench.py --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events > /tmp/prof_g.log 2>&1
f=$(find /tmp/prof_g -name "*kernel_trace.csv" | head -1); python $R/tools/gap_stats.py $f 0.6 > $R/$O/gap_stats.txt; head -3 $R/$O/gap_stats.txt; grep -A30 "gaps > 10 us" $R/$O/gap_stats.txt | head -32
