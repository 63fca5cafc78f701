#!/bin/bash
mkdir -p gpurun_out/r03_4; O=gpurun_out/r03_4
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']); print({k:(round(v['us_per_iter']),v['launches_per_iter']) for k,v in d['kernels']['per_kernel'].items()})"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; python -c "
import json; d=json.load(open('$O/bench20.json')); print(d['value'], d['ms_per_step'])"
