#!/bin/bash
# stage 2, same box, alternating: GA_DEV values given as arguments (default: product vs unet_wgrad_stream=0)
mkdir -p gpurun_out/r05s2ab
if [ $# -eq 0 ]; then set -- "" "unet_wgrad_stream=0"; fi
for rep in $(seq 1 ${REPS:-3}); do
  for v in "$@"; do
    echo -n "GA_DEV=$v  "
    GA_DEV=$v python bench.py --stage 2 --no-cpu-baseline --no-secondary --no-kernel-events --no-measure-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'it/s', round(d['ms_per_step'],3), 'ms')"
  done
done | tee gpurun_out/r05s2ab/ab.txt
