"""Times the 5x5 geometry convolutions: hand-written kernels vs MIOpen (dev tool, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gaussianavatar_amd import fused


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


x = torch.randn(1, 64, 128, 128, device="cuda", requires_grad=True)
ws = [(torch.randn(64, 64, 5, 5, device="cuda") * 0.03).requires_grad_(True) for _ in range(3)]
g = torch.randn(1, 64, 128, 128, device="cuda")


def ours():
    y = fused.geom_convs(x, ws)
    y.backward(g)


def vendor():
    y = x
    for w in ws:
        y = F.conv2d(y, w, padding=2)
    y.backward(g)


print("3 convs fwd+bwd: hand-written %7.1f us   MIOpen %7.1f us" % (timeit(ours), timeit(vendor)))
with torch.no_grad():
    print("3 convs fwd only: hand-written %7.1f us   MIOpen %7.1f us" % (
        timeit(lambda: fused.geom_convs(x, ws)), timeit(lambda: F.conv2d(F.conv2d(F.conv2d(x, ws[0], padding=2), ws[1], padding=2), ws[2], padding=2))))
