"""mlp_fwd / wgrad / bwd_data time as a function of M (fixed overhead vs per-row cost) — dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native, fused
from tools.microbench_mlp import timeit
lib = _native.ganet(); dev = torch.device("cuda"); P = fused._ptr; st = fused._stream(dev)
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
W = torch.randn(128, 128, device=dev) * 0.1; b = torch.randn(128, device=dev)
coef = torch.randn(3, 128, device=dev)
part = torch.zeros(lib.ganet_mlp_stats_floats(128), device=dev)
import os
MS = [int(os.environ['GA_M'])] if os.environ.get('GA_M') else (32768, 65536, 131072, 262144, 524288)
for M in MS:
    z = torch.randn(M, 128, device=dev); g = torch.randn(M, 128, device=dev); out = torch.empty(M, 128, device=dev)
    dW = torch.empty(128, 128, device=dev); db = torch.empty(128, device=dev)
    nb = lib.ganet_wgrad_act_workspace(M, 128, 128); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    bpart = torch.zeros(lib.ganet_mlp_bwd_data_parts() * 256, device=dev)
    t1 = timeit(lambda: fused._mlp_fwd(lib, M, 128, None, z, sc, sh, W, b, part, dev))
    t2 = timeit(lambda: lib.ganet_wgrad_act(M, 128, 128, P(g), 128, P(z), 128, P(coef), P(z), 128, P(sc), P(sh), P(dW), P(db), P(ws), nb, 0, st))
    t3 = timeit(lambda: lib.ganet_mlp_bwd_data(M, 128, P(g), 128, P(z), 128, P(coef), P(W), 128, P(out), 128, 0, P(z), 128, P(sc), P(sh), P(bpart), 0, st))
    print(f"M={M:7d}  mlp_fwd {t1:7.1f}  wgrad_act {t2:7.1f}  bwd_data(sig) {t3:7.1f} us")
