"""SSIM forward/backward kernel time on the bench shape (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import fused
from tools.microbench_mlp import timeit
a = torch.rand(2, 3, 1024, 1024, device="cuda", requires_grad=True)
b = torch.rand(2, 3, 1024, 1024, device="cuda")
def fb():
    s = fused.ssim_mean(a, b); s.backward(); a.grad = None
fused.profile_enable(["ssim_fwd", "ssim_bwd"]); fused.profile_read(True)
for _ in range(20): fb()
print({k: round(ms / n * 1e3, 1) for k, (ms, n) in fused.profile_read(True).items() if n})
