#!/bin/bash
# A/B of the decoder map path on one box: alternating runs, no kernel events (side streams on)
O=gpurun_out/r05ab; mkdir -p $O; rm -f $O/ab.txt
B="--steps 300 --warmup 30 --no-fixed-batch --no-secondary --no-cpu-baseline --no-measure-traffic --no-kernel-events"
for i in 1 2 3; do
  for v in 1 0; do
    GA_DEV=decoder_map=$v timeout 300 python bench.py $B $@ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('map=$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],3),'ms')" | tee -a $O/ab.txt
  done
done
