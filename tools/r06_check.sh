#!/bin/bash
# round 6: rasterizer parity tests, then the micro-benchmark and the size sweep (run through gpurun)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_raster_gpu.py tests/test_raster_hardening_gpu.py -x -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -25 > gpurun_out/${TAG:-r06b}_pytest_raster.txt
cat gpurun_out/${TAG:-r06b}_pytest_raster.txt
if grep -q "failed\|error" gpurun_out/${TAG:-r06b}_pytest_raster.txt; then exit 1; fi
[ -n "$TESTS_ONLY" ] || tools/r06_baseline.sh
