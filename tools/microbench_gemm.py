#!/usr/bin/env python
"""Which GEMM formulation does rocBLAS/hipBLASLt run fast for the decoder's shapes? (dev aid)"""
import time
import torch
import torch.nn.functional as F


def bench(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


M = 262144
for K, N in [(66, 128), (128, 128), (194, 128), (128, 3), (128, 1)]:
    x = torch.randn(M, K, device="cuda")
    conv = torch.nn.Conv1d(K, N, 1).cuda()
    w3, b = conv.weight, conv.bias
    w = w3.squeeze(-1)
    with torch.no_grad():
        r = {
            "linear(x,w.squeeze,b) param": bench(lambda: F.linear(x, w, b)),
            "x@w.t()+b": bench(lambda: (x @ w.t()) + b),
            "addmm(b,x,w.t())": bench(lambda: torch.addmm(b, x, w.t())),
            "linear detached clone": bench(lambda: F.linear(x, w.detach().clone(), b.detach().clone())),
        }
    xg = x.clone().requires_grad_(True)
    r["linear grad-enabled"] = bench(lambda: F.linear(xg, w, b))
    print(K, N, {k: round(v, 1) for k, v in r.items()})
