"""Times the fused decoder-layer kernels against the vendor GEMM (dev tool, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gaussianavatar_amd import _native, fused


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


def main():
    lib = _native.ganet()
    dev = torch.device("cuda")
    M = 262144
    x72 = torch.randn(M, 72, device=dev)
    z = torch.randn(M, 128, device=dev)
    sc = torch.rand(128, device=dev) + 0.5
    sh = torch.randn(128, device=dev)
    W = torch.randn(128, 128, device=dev) * 0.1
    W72 = torch.randn(128, 72, device=dev) * 0.1
    W200 = torch.randn(128, 200, device=dev) * 0.1
    W3 = torch.randn(3, 128, device=dev) * 0.1
    b = torch.randn(128, device=dev)
    part = torch.zeros(lib.ganet_mlp_stats_floats(128), device=dev)
    print("F.linear 128x128           %7.1f us" % timeit(lambda: F.linear(z, W, b)))
    print("mlp_fwd  K2=128 N=128      %7.1f us" % timeit(lambda: fused._mlp_fwd(lib, M, 128, None, z, sc, sh, W, b, part, dev)))
    print("mlp_fwd  K1=72  N=128      %7.1f us" % timeit(lambda: fused._mlp_fwd(lib, M, 128, x72, None, None, None, W72, b, part, dev)))
    print("mlp_fwd  K1=72+K2=128      %7.1f us" % timeit(lambda: fused._mlp_fwd(lib, M, 128, x72, z, sc, sh, W200, b, part, dev)))
    print("mlp_fwd  K2=128 N=3        %7.1f us" % timeit(lambda: fused._mlp_fwd(lib, M, 3, None, z, sc, sh, W3, b[:3].contiguous(), None, dev)))
    g = torch.randn(M, 128, device=dev)
    gz = torch.randn(M, 128, device=dev)
    coef = torch.randn(3, 128, device=dev)
    dW = torch.empty(128, 128, device=dev); db = torch.empty(128, device=dev)
    nb = lib.ganet_wgrad_act_workspace(M, 128, 128)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = fused._stream(dev)
    P = fused._ptr
    f = lambda: lib.ganet_wgrad_act(M, 128, 128, P(g), 128, None, 0, None, P(z), 128, P(sc), P(sh), P(dW), P(db), P(ws), nb, 0, st)
    print("wgrad_act raw g            %7.1f us" % timeit(f))
    f = lambda: lib.ganet_wgrad_act(M, 128, 128, P(g), 128, P(gz), 128, P(coef), P(z), 128, P(sc), P(sh), P(dW), P(db), P(ws), nb, 0, st)
    print("wgrad_act (G,z,coef)       %7.1f us" % timeit(f))
    out = torch.empty(M, 128, device=dev)
    part = torch.zeros(lib.ganet_mlp_bwd_data_parts() * 256, device=dev)
    f = lambda: lib.ganet_mlp_bwd_data(M, 128, P(g), 128, P(gz), 128, P(coef), P(W), 128, P(out), 128, 0, P(z), 128, P(sc), P(sh), P(part), 0, st)
    print("mlp_bwd_data sig           %7.1f us" % timeit(f))
    f = lambda: lib.ganet_mlp_bwd_data(M, 128, P(g), 128, P(gz), 128, P(coef), P(W), 128, P(out), 128, 0, None, 0, None, None, None, 0, st)
    print("mlp_bwd_data raw           %7.1f us" % timeit(f))
    bn = torch.nn.BatchNorm1d(128).cuda().train()
    print("bn+softplus fwd            %7.1f us" % timeit(lambda: fused.batchnorm_act(z, bn, "softplus")))


if __name__ == "__main__":
    main()
