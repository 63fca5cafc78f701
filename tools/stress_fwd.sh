#!/bin/bash
O=gpurun_out/stress_fwd; mkdir -p $O; : > $O/flake2.log
for i in $(seq 1 16); do timeout 120 python tools/stress_fwd.py 6 2>&1 | grep -v amdgpu.ids >> $O/flake2.log; done
grep -c "bad runs: 0" $O/flake2.log; grep -v "bad runs: 0" $O/flake2.log | head -40
