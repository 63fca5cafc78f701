#!/usr/bin/env python
"""One hidden 128->128 layer backward at M = 262,144: separate kernels (wgrad_act + mlp_bwd_data) vs ganet_mlp_bwd_fused."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
dev = torch.device("cuda"); M = 262144
torch.manual_seed(0)
G, z, sz = (torch.randn(M, 128, device=dev) for _ in range(3))
coef = torch.randn(3, 128, device=dev); W = torch.randn(128, 128, device=dev) * 0.1
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
out = torch.empty(M, 128, device=dev)
st = fused._stream(dev); P = fused._ptr
parts = max(lib.ganet_mlp_bwd_fused_parts(), lib.ganet_mlp_bwd_data_parts())
part = torch.zeros(parts * 256, device=dev)
wsb = max(lib.ganet_mlp_bwd_fused_workspace(), lib.ganet_wgrad_act_workspace(M, 128, 128))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def sep():
    _native.ganet_check(lib.ganet_wgrad_act(M, 128, 128, P(G), 128, P(z), 128, P(coef), P(sz), 128, P(sc), P(sh), None, None, ws.data_ptr(), wsb, 1, st))
    _native.ganet_check(lib.ganet_mlp_bwd_data(M, 128, P(G), 128, P(z), 128, P(coef), P(W), 128, P(out), 128, 0, P(sz), 128, P(sc), P(sh), P(part), 2, st))
def fus():
    _native.ganet_check(lib.ganet_mlp_bwd_fused(M, P(G), P(z), P(coef), P(W), 128, P(out), 0, P(sz), P(sc), P(sh), 1, P(part), ws.data_ptr(), wsb, 1, st))
for name, fn in (("separate", sep), ("fused", fus), ("separate", sep), ("fused", fus)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(name, "%.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
