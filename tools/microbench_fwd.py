#!/usr/bin/env python
"""One hidden 128->128 layer forward (ganet_mlp_fwd) at M = 262,144: time per launch and the error against float64
on the first and last 4,096 rows, plus the column sums."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
dev = torch.device("cuda"); M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
torch.manual_seed(0)
x = torch.randn(M, 128, device=dev) * 2
W = torch.randn(128, 128, device=dev) * 0.1; b = torch.randn(128, device=dev)
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev); ss = torch.randn(128, device=dev) * 0.1
z = torch.empty(M, 128, device=dev)
part = torch.zeros(lib.ganet_mlp_stats_floats(128), device=dev)
st = fused._stream(dev); P = fused._ptr
def run(order=1):
    _native.ganet_check(lib.ganet_mlp_fwd(M, 128, 0, 128, None, 0, P(x), 128, P(sc), P(sh), P(W), P(b), P(z), 128, P(part), P(ss), order, st))
for order in (1, 2):
    z.zero_(); run(order); torch.cuda.synchronize()
    ref = lambda rows: torch.nn.functional.softplus(x[rows].double() * sc.double() + sh.double()) @ W.double().t() + b.double()
    for rows in (slice(0, 4096), slice(M - 4096, M), slice(M // 2, M // 2 + 4096)):
        r = ref(rows); e = (z[rows].double() - r).abs().max() / r.abs().max()
        assert e < 2e-6, e
    full = torch.nn.functional.softplus(x * sc + sh) @ W.t() + b
    s = part.view(-1, 2, 128).double().sum(0)
    d = (full.double() - ss.double())
    e1 = (s[0] - d.sum(0)).abs().max() / d.sum(0).abs().max(); e2 = (s[1] - (d * d).sum(0)).abs().max() / (d * d).sum(0).abs().max()
    print("order %d: max rel err vs float64 %.2e | column sums rel err %.2e %.2e | z vs fp32 torch %.2e" % (
        order, float(e), float(e1), float(e2), float((z - full).abs().max())))
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): run(1 + (i & 1))
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50 * 1e6
fused.profile_enable(["mlp_fwd"]); fused.profile_read(True)
for i in range(50): run(1 + (i & 1))
torch.cuda.synchronize(); r = fused.profile_read(True)["mlp_fwd"]
print("%.1f us per launch (HIP events around the kernel), %.1f us back to back" % (r[0] / r[1] * 1e3, wall))
