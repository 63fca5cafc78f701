#!/usr/bin/env python
"""Phase stamps of conv5_kernel (needs a -DGANET_CONV_TRACE build, GA_DEV=lib_dir=<dir>): cycles per phase of waves 0 (k-quarter 0)
and 7 (k-quarter 3) of block 0, and the start / end spread of the 256 workgroups."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
x = torch.randn(1, 64, 128, 128, device="cuda")
ws = [torch.randn(64, 64, 5, 5, device="cuda") * 0.03 for _ in range(3)]
with torch.no_grad():
    for _ in range(5): y = fused.geom_convs(x, ws)
torch.cuda.synchronize()
tr = np.zeros((2, 8), dtype=np.uint64); bl = np.zeros((1024, 2), dtype=np.uint64)
lib.ganet_dev_conv_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.ganet_dev_conv_trace(tr.ctypes.data, bl.ctypes.data) == 0
t = tr.astype(np.int64)
names = ["B + halo loads issued", "halo split + LDS writes", "barrier", "tap loop (300 MFMAs)", "barrier", "hand-over writes + barrier", "add + store"]
for w, lab in ((0, "wave 0 (k-quarter 0)"), (1, "wave 7 (k-quarter 3)")):
    print(lab, " | ".join("%s %d" % (n, t[w, i + 1] - t[w, i]) for i, n in enumerate(names)), "| total", t[w, 7] - t[w, 0])
b = bl[:256].astype(np.int64); t0 = b[:, 0].min(); u = (b - t0) / 100.0
print("workgroups (us): start max %.1f | duration median %.1f min %.1f max %.1f | end max %.1f" % (
    u[:, 0].max(), np.median(u[:, 1] - u[:, 0]), (u[:, 1] - u[:, 0]).min(), (u[:, 1] - u[:, 0]).max(), u[:, 1].max()))

# the weight gradient (conv5_wgrad_kernel): 47 chunks x 5 kernel rows = 235 workgroups
g = torch.randn(1, 64, 128, 128, device="cuda")
xs = [x.clone().requires_grad_(True)]
wsr = [w.clone().requires_grad_(True) for w in ws]
for _ in range(3):
    y = fused.geom_convs(xs[0], wsr); y.backward(g)
torch.cuda.synchronize()
tr = np.zeros((2, 8), dtype=np.uint64); bl = np.zeros((1024, 2), dtype=np.uint64)
lib.ganet_dev_wgrad_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.ganet_dev_wgrad_trace(tr.ctypes.data, bl.ctypes.data) == 0
t = tr.astype(np.int64)
names = ["first loads issued", "run loop (11 runs x 30 MFMAs)", "hand-over writes", "barrier", "add + store partial tile"]
for w, lab in ((0, "wgrad wave 0 (group 0)"), (1, "wgrad wave 7 (group 1)")):
    print(lab, " | ".join("%s %d" % (n, t[w, i + 1] - t[w, i]) for i, n in enumerate(names)), "| total", t[w, 5] - t[w, 0])
b = bl[:235].astype(np.int64); t0 = b[:, 0].min(); u = (b - t0) / 100.0
print("wgrad workgroups (us): start max %.1f | duration median %.1f min %.1f max %.1f | end max %.1f" % (
    u[:, 0].max(), np.median(u[:, 1] - u[:, 0]), (u[:, 1] - u[:, 0]).min(), (u[:, 1] - u[:, 0]).max(), u[:, 1].max()))
