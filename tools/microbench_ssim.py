#!/usr/bin/env python
"""SSIM + L1 forward / backward (ganet_ssim_fwd / ganet_ssim_bwd) at the headline shape: 2 frames x 3 planes x 1024^2.
Prints the time per launch and the error against the reference's formulation (five grouped convolutions) in torch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gaussianavatar_amd import fused

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else H
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda")
torch.manual_seed(0)
a = torch.rand(B, 3, H, W, device=dev).requires_grad_()
b = (a.detach() + 0.1 * torch.randn_like(a)).clamp(0, 1)


def torch_ssim_l1(x, y):
    g = torch.exp(-(torch.arange(11, device=x.device, dtype=x.dtype) - 5) ** 2 / (2 * 1.5 ** 2)); g = g / g.sum()
    w = (g[:, None] * g[None, :]).expand(3, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t, w, padding=5, groups=3)
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return m.mean(), (x - y).abs().mean()


s, l = fused.ssim_l1_mean(a, b)
(0.2 * (1 - s) + 0.8 * l).backward()
g = a.grad.clone(); a.grad = None
ad = a.detach().double().requires_grad_()
sr, lr = torch_ssim_l1(ad, b.double())
(0.2 * (1 - sr) + 0.8 * lr).backward()
print("ssim %.8f ref %.8f | l1 %.8f ref %.8f | grad max err %.3e (ref max %.3e)" % (
    float(s), float(sr), float(l), float(lr), float((g.double() - ad.grad).abs().max()), float(ad.grad.abs().max())))
for name in ("fwd", "bwd"):
    fused.profile_enable(["ssim_" + name]); fused.profile_read(True)
    for _ in range(20):
        s, l = fused.ssim_l1_mean(a, b); (s + l).backward(); a.grad = None
    torch.cuda.synchronize()
    r = fused.profile_read(True)
    print(name, {k: round(v[0] / v[1] * 1e3, 1) for k, v in r.items() if v[1]})
fused.profile_enable([])
