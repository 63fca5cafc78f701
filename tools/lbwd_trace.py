#!/usr/bin/env python
"""Phase time stamps of ganet_layer_bwd's consumer wave 0 / producer wave 4 of block 0 (needs a -DGANET_LBWD_TRACE build:
GA_DEV=lib_dir=build_v_trace). Prints cycles per phase, averaged over rounds 4..27."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
dev = torch.device("cuda"); M = 262144
torch.manual_seed(0)
G, z, sz = (torch.randn(M, 128, device=dev) for _ in range(3))
coef = torch.randn(3, 128, device=dev); W = torch.randn(128, 128, device=dev) * 0.1
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
out = torch.empty(M, 128, device=dev)
st = fused._stream(dev); P = fused._ptr
part = torch.zeros(lib.ganet_mlp_bwd_fused_parts() * 256, device=dev)
wsb = lib.ganet_mlp_bwd_fused_workspace(); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
for _ in range(5):
    _native.ganet_check(lib.ganet_mlp_bwd_fused(M, P(G), P(z), P(coef), P(W), 128, P(out), 0, P(sz), P(sc), P(sh), 1, P(part), ws.data_ptr(), wsb, 1, st))
torch.cuda.synchronize()
buf = np.zeros((2, 64, 16), dtype=np.uint64)
lib.ganet_dev_lbwd_trace.argtypes = [ctypes.c_void_p]
assert lib.ganet_dev_lbwd_trace(buf.ctypes.data) == 0
t = buf.astype(np.int64)
c, p = t[0], t[1]
R = slice(4, 28)
print("consumer (cycles): dgrad %.0f  wgrad %.0f  epilogue %.0f  barrier wait %.0f  | round %.0f" % (
    (c[R, 1] - c[R, 0]).mean(), (c[R, 2] - c[R, 1]).mean(), (c[R, 3] - c[R, 2]).mean(), (c[R, 4] - c[R, 3]).mean(),
    (c[5:29, 0] - c[4:28, 0]).mean()))
print("producer (cycles): wait loads %.0f  produce0 %.0f  issue loads0 %.0f  produce1 %.0f  issue loads1 %.0f  barrier wait %.0f | round %.0f" % (
    (p[R, 1] - p[R, 0]).mean(), (p[R, 2] - p[R, 1]).mean(), (p[R, 3] - p[R, 2]).mean(), (p[R, 4] - p[R, 3]).mean(),
    (p[R, 5] - p[R, 4]).mean(), (p[R, 6] - p[R, 5]).mean(), (p[5:29, 0] - p[4:28, 0]).mean()))
print("first rounds, consumer stamps relative:", (c[:6, :5] - c[0, 0]).tolist())
print("first rounds, producer stamps relative:", (p[:6, :7] - c[0, 0]).tolist())
print("whole kernel, consumer: prologue %d  loop %d  tail %d | producer: prologue %d loop %d tail %d" % (
    c[63, 1] - c[63, 0], c[63, 2] - c[63, 1], c[63, 3] - c[63, 2], p[63, 1] - p[63, 0], p[63, 2] - p[63, 1], p[63, 3] - p[63, 2]))
print("rounds (consumer, start to start):", (c[1:32, 0] - c[0:31, 0]).tolist())
bl = np.zeros((256, 4), dtype=np.uint64)
lib.ganet_dev_lbwd_blocks.argtypes = [ctypes.c_void_p]
assert lib.ganet_dev_lbwd_blocks(bl.ctypes.data) == 0
b = bl.astype(np.int64); t0 = b[:, 0].min(); u = (b - t0) / 100.0
print("workgroups (us): start max %.1f | prologue median %.1f max %.1f | loop median %.1f min %.1f max %.1f | loop end: min %.1f median %.1f max %.1f | end max %.1f" % (
    u[:, 0].max(), np.median(u[:, 1] - u[:, 0]), (u[:, 1] - u[:, 0]).max(), np.median(u[:, 2] - u[:, 1]), (u[:, 2] - u[:, 1]).min(),
    (u[:, 2] - u[:, 1]).max(), u[:, 2].min(), np.median(u[:, 2]), u[:, 2].max(), u[:, 3].max()))
print("loop time by XCD (block %% 8):", [round(float(np.median((u[:, 2] - u[:, 1])[x::8])), 1) for x in range(8)])
