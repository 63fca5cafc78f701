"""Split-bf16 decoder kernels vs the fp32-MFMA kernels: error against float64 and time (dev tool, GPU box)."""
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


def err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


def main():
    lib = _native.ganet()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    M = int(os.environ.get("M", 262144))
    P, st = fused._ptr, fused._stream(dev)
    x72 = torch.randn(M, 72, device=dev); x72[:, 66:] = 0
    z = torch.randn(M, 128, device=dev) * 1.5 + 0.3
    sc = torch.rand(128, device=dev) + 0.5
    sh = torch.randn(128, device=dev)
    W = torch.randn(128, 128, device=dev) * 0.1
    W72 = torch.randn(128, 72, device=dev) * 0.1
    W200 = torch.randn(128, 200, device=dev) * 0.1
    W3 = torch.randn(3, 128, device=dev) * 0.1
    b = torch.randn(128, device=dev)
    g = torch.randn(M, 128, device=dev)
    gz = torch.randn(M, 128, device=dev)
    coef = torch.randn(3, 128, device=dev)
    act64 = F.softplus(z.double() * sc.double() + sh.double())
    dz64 = g.double() * coef[0].double() + gz.double() * coef[1].double() + coef[2].double()
    part = torch.zeros(lib.ganet_mlp_stats_floats(128), device=dev)
    bpart = torch.zeros(lib.ganet_mlp_bwd_data_parts() * 256, device=dev)
    nb = lib.ganet_wgrad_act_workspace(M, 128, 128)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dW = torch.empty(128, 128, device=dev); db = torch.empty(128, device=dev)
    dW72 = torch.empty(128, 72, device=dev)
    out = torch.empty(M, 128, device=dev)

    cases = {
        "fwd K2=128": (lambda: fused._mlp_fwd(lib, M, 128, None, z, sc, sh, W, b, part, dev),
                       lambda: act64 @ W.double().t() + b.double()),
        "fwd K1=72": (lambda: fused._mlp_fwd(lib, M, 128, x72, None, None, None, W72, b, part, dev),
                      lambda: x72.double() @ W72.double().t() + b.double()),
        "fwd K1=72+K2=128": (lambda: fused._mlp_fwd(lib, M, 128, x72, z, sc, sh, W200, b, part, dev),
                             lambda: torch.cat([x72.double(), act64], 1) @ W200.double().t() + b.double()),
        "fwd head N=3": (lambda: fused._mlp_fwd(lib, M, 3, None, z, sc, sh, W3, b[:3].contiguous(), None, dev),
                         lambda: act64 @ W3.double().t() + b[:3].double()),
    }

    def bwd(sig, accumulate=False, O=128, Wm=None):
        Wm = W if Wm is None else Wm
        def f():
            if accumulate and not os.environ.get("NOCAT"):
                out.fill_(1.0)
            _native.ganet_check(lib.ganet_mlp_bwd_data(
                M, O, P(g), 128, P(gz), 128, P(coef), P(Wm), Wm.stride(0), P(out), 128, int(accumulate),
                P(z) if sig else None, 128 if sig else 0, P(sc) if sig else None, P(sh) if sig else None,
                P(bpart) if sig else None, 0, st))
            return out[:, :O]
        def ref():
            r = dz64 @ Wm.double()
            if accumulate:
                r = r + 1.0
            if sig:
                r = r * torch.sigmoid(z.double() * sc.double() + sh.double())[:, :O]
            return r
        return f, ref
    cases["bwd sig"] = bwd(True)
    cases["bwd raw"] = bwd(False)
    cases["bwd acc"] = bwd(False, True)
    cases["bwd acc sig"] = bwd(True, True)
    cases["bwd O=66"] = bwd(False, False, 66, W200[:, :66])

    def wg(gpro, K=128):
        def f():
            xx, scx, shx, d = (z, sc, sh, dW) if K == 128 else (x72, None, None, dW72)
            _native.ganet_check(lib.ganet_wgrad_act(
                M, 128, K, P(g), 128, P(gz) if gpro else None, 128 if gpro else 0, P(coef) if gpro else None,
                P(xx), K, P(scx), P(shx), P(d), P(db), P(ws), nb, 0, st))
            return d if os.environ.get("NOCAT") else torch.cat([d.reshape(-1), db])
        def ref():
            gg = dz64 if gpro else g.double()
            xx = act64 if K == 128 else x72.double()
            return torch.cat([(gg.t() @ xx).reshape(-1), gg.sum(0)])
        return f, ref
    cases["wgrad (G,z,coef)"] = wg(True)
    cases["wgrad raw g"] = wg(False)
    cases["wgrad K=72 gpro"] = wg(True, 72)

    only = os.environ.get("ONLY")
    for name, (f, ref) in cases.items():
        if only and only not in name:
            continue
        r64 = ref()
        row = "%-20s" % name
        for mode in (0, 1):
            lib.ganet_set_mfma_mode(mode)
            e = err(f().clone(), r64)
            os.environ["NOCAT"] = "1"
            t = timeit(f)
            del os.environ["NOCAT"]
            row += "  %s: err %.2e  %7.1f us" % ("f32  " if mode == 0 else "split", e, t)
        print(row, flush=True)
    lib.ganet_set_mfma_mode(1)


if __name__ == "__main__":
    main()
