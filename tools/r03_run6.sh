#!/bin/bash
for i in 1 2; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('bench native', d['value'], d['ms_per_step'])"; done
python -c "
import sys; sys.argv=['bench.py','--steps','200','--warmup','30','--no-cpu-baseline']
import gaussianavatar_amd.fused as f; f._NATIVE_DECODER=False
import bench; bench.main()" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('bench python-seq', d['value'], d['ms_per_step'])"
GA_WGRAD_STREAM=0 python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('bench native no-side', d['value'], d['ms_per_step'])"
