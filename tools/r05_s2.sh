#!/bin/bash
O=gpurun_out/r05s2; mkdir -p $O
timeout 800 python -m pytest tests/test_unet_gpu.py -q -m gpu 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fused_gpu.py -q -m gpu -k "stage2 or production or whole_net" 2>&1 | tail -5
B="--steps 100 --warmup 20 --no-fixed-batch --no-secondary --no-cpu-baseline --no-measure-traffic --stage 2"
for v in 1 0 1 0; do
GA_DEV=native_unet=$v timeout 300 python bench.py $B --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('native_unet=$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],3),'ms')"
done
