#!/bin/bash
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q -k "fused_input or production or bwd_fused or fused_decoder" 2>&1 | grep -E "Error|error|assert|passed|failed|fault" | cut -c1-250 | head -20
python tools/microbench_bwd_fused_input.py 2>&1 | grep -v amdgpu
bash tools/kstats.sh layer_bwd wgrad_act mlp_bwd 2>&1 | tail -12
