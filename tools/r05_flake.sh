#!/bin/bash
O=gpurun_out/r05b; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 300 python -m pytest tests/test_decoder_map_gpu.py -q -m gpu -k "mlp_fwd_add or upsample_z" 2>&1 | grep -E "^FAILED|passed|failed|AssertionError:|elements beyond" | cut -c1-700 >> $O/flake.log
done
cat $O/flake.log
