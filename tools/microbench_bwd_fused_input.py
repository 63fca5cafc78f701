#!/usr/bin/env python
"""Backward of a layer fed by the raw decoder input (conv1 / conv5's input half) at M = 262,144: separate kernels
(wgrad_act K = 72 + mlp_bwd_data O = 66) vs ganet_mlp_bwd_fused_input, with and without accumulation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
dev = torch.device("cuda"); M = 262144
torch.manual_seed(0)
G, z = (torch.randn(M, 128, device=dev) for _ in range(2))
x = torch.randn(M, 72, device=dev); x[:, 66:] = 0
coef = torch.randn(3, 128, device=dev); W = torch.randn(128, 66, device=dev) * 0.1
out = torch.zeros(M, 72, device=dev)
st = fused._stream(dev); P = fused._ptr
wsb = max(lib.ganet_mlp_bwd_fused_workspace(), lib.ganet_wgrad_act_workspace(M, 128, 128))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def sep(acc):
    _native.ganet_check(lib.ganet_wgrad_act(M, 128, 72, P(G), 128, P(z), 128, P(coef), P(x), 72, None, None, None, None, ws.data_ptr(), wsb, 1, st))
    _native.ganet_check(lib.ganet_mlp_bwd_data(M, 66, P(G), 128, P(z), 128, P(coef), P(W), 66, P(out), 72, acc, None, 0, None, None, None, 2, st))
def fus(acc):
    _native.ganet_check(lib.ganet_mlp_bwd_fused_input(M, P(G), P(z), P(coef), P(W), 66, 66, P(out), 72, acc, P(x), ws.data_ptr(), wsb, 1, st))
for name, fn, acc in (("separate", sep, 0), ("fused", fus, 0), ("separate acc", sep, 1), ("fused acc", fus, 1), ("fused", fus, 0)):
    for _ in range(5): fn(acc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn(acc)
    torch.cuda.synchronize(); print(name, "%.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
