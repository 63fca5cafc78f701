#!/usr/bin/env python
"""Phase time stamps of ganet_layer_fwd's consumer wave 0 / producer wave 4 of block 0 (needs a -DGANET_LFWD_TRACE build:
GA_DEV=lib_dir=<dir>). Prints cycles per phase, averaged over rounds 4..27."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
dev = torch.device("cuda"); M = 262144
torch.manual_seed(0)
x = torch.randn(M, 128, device=dev); W = torch.randn(128, 128, device=dev) * 0.1; b = torch.randn(128, device=dev)
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
z = torch.empty(M, 128, device=dev); part = torch.zeros(lib.ganet_mlp_stats_floats(128), device=dev)
st = fused._stream(dev); P = fused._ptr
if os.environ.get("THREE"):     # the three-branch launch (block 0 = branch 0 of group 0)
    W = torch.randn(3, 128, 128, device=dev) * 0.1; b = torch.randn(3, 128, device=dev)
    z = torch.empty(3, M, 128, device=dev); part = torch.zeros(3, 256, 256, device=dev)
    lib.ganet_dev_layer_fwd3.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 8
for _ in range(5):
    if os.environ.get("THREE"):
        _native.ganet_check(lib.ganet_dev_layer_fwd3(M, P(x), P(sc), P(sh), P(W), P(b), P(z), P(part), st))
    else:
        _native.ganet_check(lib.ganet_mlp_fwd(M, 128, 0, 128, None, 0, P(x), 128, P(sc), P(sh), P(W), P(b), P(z), 128, P(part), None, 1, st))
torch.cuda.synchronize()
buf = np.zeros((2, 64, 8), dtype=np.uint64)
lib.ganet_dev_lfwd_trace.argtypes = [ctypes.c_void_p]
assert lib.ganet_dev_lfwd_trace(buf.ctypes.data) == 0
t = buf.astype(np.int64); c, p = t[0], t[1]
R = slice(4, 28)
print("consumer (cycles): mfma %.0f  epilogue %.0f  barrier wait %.0f | round %.0f" % (
    (c[R, 1] - c[R, 0]).mean(), (c[R, 2] - c[R, 1]).mean(), (c[R, 3] - c[R, 2]).mean(), (c[5:29, 0] - c[4:28, 0]).mean()))
print("producer (cycles): drain %.0f  wait loads %.0f  produce + loads %.0f  barrier wait %.0f | round %.0f" % (
    (p[R, 1] - p[R, 0]).mean(), (p[R, 2] - p[R, 1]).mean(), (p[R, 3] - p[R, 2]).mean(), (p[R, 4] - p[R, 3]).mean(),
    (p[5:29, 0] - p[4:28, 0]).mean()))
print("rounds (consumer, start to start):", (c[1:32, 0] - c[0:31, 0]).tolist())
print("producer wait-loads per round:", (p[0:32, 2] - p[0:32, 1]).tolist())
bl = np.zeros((256, 4), dtype=np.uint64)
lib.ganet_dev_lfwd_blocks.argtypes = [ctypes.c_void_p]
assert lib.ganet_dev_lfwd_blocks(bl.ctypes.data) == 0
b = bl.astype(np.int64); t0 = b[:, 0].min(); u = (b - t0) / 100.0
print("workgroups (us): start max %.1f | prologue median %.1f max %.1f | loop median %.1f min %.1f max %.1f | loop end: min %.1f median %.1f max %.1f" % (
    u[:, 0].max(), np.median(u[:, 1] - u[:, 0]), (u[:, 1] - u[:, 0]).max(), np.median(u[:, 2] - u[:, 1]), (u[:, 2] - u[:, 1]).min(),
    (u[:, 2] - u[:, 1]).max(), u[:, 2].min(), np.median(u[:, 2]), u[:, 2].max()))
print("loop time by XCD (block %% 8):", [round(float(np.median((u[:, 2] - u[:, 1])[x::8])), 1) for x in range(8)])
