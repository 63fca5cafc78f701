#!/usr/bin/env python
"""development: phase stamps of one chunk-sort workgroup (build with tools/build_gsr_variant.sh <dir> -DGSR_SORT_TRACE=<rank>,
run with GA_DEV=lib_dir=<dir>): where a 2048-key chunk sort spends its time inside the launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native
import tools.bench_raster as br
kind = sys.argv[1] if len(sys.argv) > 1 else "avatar_3mm"
N, W, H = br.SIZES["200k"]
m, bt, pts = br.body(N, W, H)
it, _ = br.make_iteration(kind, m, bt, pts, br.gaussian_set(kind, N, 0), W, H)
for _ in range(6):
    it()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib = _native.gsr()
lib.gsr_dev_sort_trace(buf)
t = list(buf)
names = {0: "kernel entry", 1: "tile span loaded", 2: "keys in LDS", 3: "8-key networks done", 20: "sorted", 21: "stored"}
print("keys in this chunk:", t[30])
prev = t[0]
for i in [0, 1, 2, 3] + list(range(4, 16)) + [20, 21]:
    if t[i] == 0:
        continue
    print(f"  {names.get(i, 'merge level %d done' % (i - 4)):24s} +{(t[i] - prev) * 10:7d} ns   (at {(t[i] - t[0]) * 10} ns)")
    prev = t[i]
