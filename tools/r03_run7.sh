#!/bin/bash
O=gpurun_out/r03_7; mkdir -p $O
timeout 1500 python -m pytest tests/test_raster_hardening_gpu.py -q 2>&1 | tail -40 | tee $O/pytest_hard.txt
