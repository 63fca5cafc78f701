#!/bin/bash
# last call of round 5: full GPU suite, smoke, the driver's bench command, stage-2 line + kernel table -> gpurun_out/r05_last/
O=gpurun_out/r05_last; mkdir -p $O
bash tools/run_gpu_checks.sh r05_last
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 60 --stage 2 --no-fixed-batch > $O/stage2.json 2> $O/stage2.err
python -c "import json; d=json.load(open('$O/stage2.json')); print('stage2', round(d['value'],1), round(d['ms_per_step'],3))"
bash tools/kstats_s2.sh > $O/stage2_kernel_stats.txt 2>&1; tail -1 $O/stage2_kernel_stats.txt
bash tools/unet_calls.sh > /dev/null 2>&1; cp gpurun_out/r05unet/calls.txt $O/unet_calls.txt
