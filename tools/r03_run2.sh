#!/bin/bash
mkdir -p gpurun_out/r03_2; O=gpurun_out/r03_2
timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -k "bwd_fused or fused_decoder" 2>&1 | tail -3 | tee $O/pytest_fused.txt
for v in "" build_v_sym; do
  echo "--- microbench variant '$v'"; GA_LIB_DIR=${v:+$PWD/$v} timeout 300 python tools/microbench_bwd_fused.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_$v.txt
done
bash tools/pmc_lbwd.sh > $O/pmc_lbwd.txt 2>&1; grep -A9 "layer_bwd" $O/pmc_lbwd.txt | head -60
echo "--- bench"; timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step']); print({k:(round(v['us_per_iter']),v['launches_per_iter']) for k,v in d['kernels']['per_kernel'].items()})"
