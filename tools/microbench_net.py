#!/usr/bin/env python
"""Micro-benchmarks of the torch-side pieces of the hot path on the GPU box (development aid):
GEMM formulations for the decoder, 5x5 conv formulations, BN/softplus, SSIM convs."""
import sys
import time

import torch
import torch.nn.functional as F


def bench(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    dev = "cuda"
    M, K, N = 262144, 128, 128
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    xT = x.t().contiguous()          # [K, M]
    flops = 2.0 * M * K * N
    res = {}
    res["linear(x,w,b)        [M,K]x[N,K]^T"] = bench(lambda: F.linear(x, w, b))
    res["x @ w.t()            "] = bench(lambda: x @ w.t())
    wt = w.t().contiguous()
    res["x @ wt (contig)      "] = bench(lambda: x @ wt)
    res["w @ xT               [N,K]x[K,M]"] = bench(lambda: w @ xT)
    res["conv1d [1,K,M]       "] = bench(lambda: F.conv1d(xT[None], w[:, :, None], b))
    x3 = x.view(512, 512, K)
    res["bmm 512x[512,K]x[K,N]"] = bench(lambda: torch.matmul(x3, wt))
    g = torch.randn(M, N, device=dev)
    res["grad_in  g @ w       [M,N]x[N,K]"] = bench(lambda: g @ w)
    res["grad_w   g.t() @ x   [N,M]x[M,K]"] = bench(lambda: g.t() @ x)
    gT = g.t().contiguous()
    res["grad_w   gT @ x (contig)"] = bench(lambda: gT @ x)
    for k, v in res.items():
        print(f"{k:45s} {v:9.1f} us  {flops / v / 1e6:7.1f} TF/s")
    # bf16 for reference only (NOT used: the reference computes in fp32)
    xb, wb = x.bfloat16(), w.bfloat16()
    v = bench(lambda: F.linear(xb, wb))
    print(f"{'[info] bf16 linear':45s} {v:9.1f} us  {flops / v / 1e6:7.1f} TF/s")

    # BN + softplus over [M,128]
    y = torch.randn(M, N, device=dev, requires_grad=True)
    bn = torch.nn.BatchNorm1d(N).to(dev)
    print(f"{'BN1d fwd [M,128]':45s} {bench(lambda: bn(y)):9.1f} us")
    print(f"{'softplus fwd':45s} {bench(lambda: F.softplus(y)):9.1f} us")

    def bn_sp_fb():
        o = F.softplus(bn(y))
        o.backward(g)
        y.grad = None
    print(f"{'BN+softplus fwd+bwd':45s} {bench(bn_sp_fb):9.1f} us")

    # 5x5 conv 64->64 on 128^2
    xi = torch.randn(1, 64, 128, 128, device=dev, requires_grad=True)
    wc = torch.randn(64, 64, 5, 5, device=dev, requires_grad=True)
    go = torch.randn(1, 64, 128, 128, device=dev)

    def conv_fb():
        o = F.conv2d(xi, wc, padding=2)
        o.backward(go)
        xi.grad = None
        wc.grad = None
    for bm in (False, True):
        torch.backends.cudnn.benchmark = bm
        print(f"{'conv5x5 fwd cudnn.benchmark=' + str(bm):45s} {bench(lambda: F.conv2d(xi, wc, padding=2)):9.1f} us")
        print(f"{'conv5x5 fwd+bwd cudnn.benchmark=' + str(bm):45s} {bench(conv_fb):9.1f} us")

    def conv_unfold():
        cols = F.unfold(xi, 5, padding=2)                      # [1, 1600, 16384]
        return (wc.view(64, -1) @ cols[0]).view(1, 64, 128, 128)

    def conv_unfold_fb():
        o = conv_unfold()
        o.backward(go)
        xi.grad = None
        wc.grad = None
    print(f"{'conv5x5 unfold+gemm fwd':45s} {bench(conv_unfold):9.1f} us")
    print(f"{'conv5x5 unfold+gemm fwd+bwd':45s} {bench(conv_unfold_fb):9.1f} us")

    # SSIM style depthwise separable conv on [2,15,1024,1024]
    s = torch.randn(2, 15, 1024, 1024, device=dev, requires_grad=True)
    kx = torch.randn(15, 1, 1, 11, device=dev)
    ky = torch.randn(15, 1, 11, 1, device=dev)
    f2 = lambda: F.conv2d(F.conv2d(s, kx, padding=(0, 5), groups=15), ky, padding=(5, 0), groups=15)
    print(f"{'ssim separable depthwise fwd':45s} {bench(f2):9.1f} us")
    gs = torch.randn(2, 15, 1024, 1024, device=dev)

    def f2_fb():
        o = f2()
        o.backward(gs)
        s.grad = None
    print(f"{'ssim separable depthwise fwd+bwd':45s} {bench(f2_fb):9.1f} us")
    # grid_sample
    feat = torch.randn(1, 64, 128, 128, device=dev, requires_grad=True)
    idx = torch.stack(torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij"), -1).float().to(dev) / 511
    grid = (idx.reshape(1, 512, 512, 2) * 2 - 1).transpose(1, 2)
    gg = torch.randn(1, 64, 512, 512, device=dev)

    def gs_fb():
        o = F.grid_sample(feat, grid, mode="bilinear", align_corners=False)
        o.backward(gg)
        feat.grad = None
    print(f"{'grid_sample fwd+bwd':45s} {bench(gs_fb):9.1f} us")


if __name__ == "__main__":
    main()
