import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianavatar_amd import fused
M = 262144
for N, K in [(128, 128), (128, 66), (128, 194), (3, 128)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda", requires_grad=True)
    b = torch.randn(N, device="cuda", requires_grad=True); g = torch.randn(M, N, device="cuda")
    y = fused.linear(x, w, b)
    def f():
        w.grad = None; b.grad = None
        y.backward(g, retain_graph=True)
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30 * 1e6
    print(f"N={N} K={K}: wgrad (incl. reduce) {dt:.1f} us  -> {2.0*M*N*K/dt/1e6:.1f} TF/s")
