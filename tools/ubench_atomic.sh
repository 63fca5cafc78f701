#!/bin/bash
# builds and runs tools/ubench/atomic_rate.hip -> <outdir>/atomic_rate.txt
O=${1:-gpurun_out}; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_rate.hip -o /tmp/atomic_rate 2>/dev/null && /tmp/atomic_rate > $O/atomic_rate.txt 2>&1; head -8 $O/atomic_rate.txt
