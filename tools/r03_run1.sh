#!/bin/bash
# round-3 GPU batch 1: transposing-read probe, fused layer backward parity + timing (both wave orders), bench A/B
mkdir -p gpurun_out/r03_1; O=gpurun_out/r03_1
tools/ubench/tr16_probe > $O/tr16.txt 2>&1; tail -2 $O/tr16.txt
timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -k "bwd_fused or fused_decoder or production" 2>&1 | tail -15 | tee $O/pytest_fused.txt
echo "--- microbench order1 (default)"; timeout 300 python tools/microbench_bwd_fused.py 2>&1 | tee $O/mb_order1.txt
echo "--- microbench order0"; GA_LIB_DIR=$PWD/build_v_order0 timeout 300 python tools/microbench_bwd_fused.py 2>&1 | tee $O/mb_order0.txt
echo "--- bench fused"; timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; python -c "
import json; d=json.load(open('$O/bench_fused.json')); print(d['value'], d['ms_per_step']); print({k:(round(v['us_per_iter']),v['launches_per_iter']) for k,v in d['kernels']['per_kernel'].items()})"
echo "--- bench separate"; timeout 600 python -c "
import sys; sys.argv=['bench.py','--steps','100','--warmup','20','--no-cpu-baseline']
import gaussianavatar_amd.fused as f; f._FUSED_BWD=False
import bench; bench.main()" > $O/bench_sep.json 2> $O/bench_sep.err; python -c "
import json; d=json.load(open('$O/bench_sep.json')); print(d['value'], d['ms_per_step']); print({k:(round(v['us_per_iter']),v['launches_per_iter']) for k,v in d['kernels']['per_kernel'].items()})"
