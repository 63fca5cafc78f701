#!/bin/bash
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q 2>&1 | tail -2
bash tools/kstats.sh tile_ scatter preprocess render 2>&1 | tail -14
