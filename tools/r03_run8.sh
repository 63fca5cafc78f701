#!/bin/bash
O=gpurun_out/r03_8; mkdir -p $O
timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_raster_hardening_gpu.py -x -q 2>&1 | tail -15 | tee $O/pytest_raster.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step']); print({k:(round(v['us_per_iter']),v['launches_per_iter']) for k,v in d['kernels']['per_kernel'].items() if k in ('preprocess','tile_scan','scatter','tile_sort','render_fwd','render_bwd','preprocess_bwd')})"; tail -3 $O/bench.err
