#!/bin/bash
mkdir -p gpurun_out/r03_3; O=gpurun_out/r03_3
timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -k "bwd_fused or fused_decoder" 2>&1 | tail -3
timeout 300 python tools/microbench_bwd_fused.py 2>&1 | grep -v amdgpu | tee $O/mb.txt
GA_LIB_DIR=$PWD/build_v_trace timeout 300 python tools/lbwd_trace.py 2>&1 | grep -v amdgpu | tee $O/trace.txt
