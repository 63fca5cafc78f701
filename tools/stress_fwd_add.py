"""development: repeat ganet_mlp_fwd_add at M = 262,144 and describe any mismatch against float64 (which slabs / rows / columns)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from gaussianavatar_amd import _native
from tests.test_decoder_map_gpu import _grid_setup, _upsample64, _ptr, _stream

lib = _native.ganet()
b, feat, S = 1, 128, 512
torch.manual_seed(0)
net, uv, mats, taps, grid = _grid_setup(b, feat, S)
M = b * S * S
P = torch.randn(b * feat * feat, 256, device="cuda")
x2 = torch.randn(M, 128, device="cuda") * 2
sc = torch.rand(128, device="cuda") + 0.5
sh = torch.randn(128, device="cuda") * 0.3
W = torch.randn(128, 128, device="cuda") * 0.1
bias = torch.randn(128, device="cuda")
Wuv = torch.randn(128, 2, device="cuda")
a = F.softplus(x2.double() * sc.double() + sh.double())
base = a @ W.double().t() + bias.double()
addend = _upsample64(mats, P[:, 128:].double(), b, feat, S) + uv.reshape(M, 2).double() @ Wuv.double().t()
ref = base + addend
Pv = P[:, 128:]
bad_runs = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for order in (0, 2):
        z = torch.full((M, 128), float("nan"), device="cuda")
        _native.ganet_check(lib.ganet_mlp_fwd_add(ctypes.byref(grid), _ptr(x2), _ptr(sc), _ptr(sh), _ptr(W), _ptr(bias),
                                                  _ptr(Pv), 256, _ptr(Wuv), _ptr(z), None, None, order, _stream()))
        err = (z.double() - ref).abs()
        nb = int((~(err <= 1e-4)).sum())
        if nb:
            bad_runs += 1
            idx = torch.nonzero(~(err <= 1e-4))
            rows, cols = idx[:, 0], idx[:, 1]
            slabs = torch.unique(rows // 32)
            e2 = (z.double() - base).abs()           # is the bad value "no addend"?
            noadd = int(((e2 <= 1e-4) & ~(err <= 1e-4)).sum())
            print(f"iter {it} order {order}: {nb} bad elements, {slabs.numel()} slabs (first {slabs[:8].tolist()}), "
                  f"rows-in-slab {torch.unique(rows % 32)[:16].tolist()}, cols {torch.unique(cols)[:16].tolist()} "
                  f"n_cols {torch.unique(cols).numel()}, max err {float(err[~torch.isnan(err)].max()) if nb else 0:.3f}, "
                  f"nan {int(torch.isnan(z).sum())}, equal-to-no-addend {noadd}", flush=True)
            # is it the addend of another slab? compare with addend of slab +-256 etc.
            r0 = int(rows[0]); c0 = int(cols[0])
            got_add = float(z[r0, c0].double() - base[r0, c0])
            cands = {d: float(addend[r0 + d * 32, c0]) for d in (-512, -256, -1, 0, 1, 256, 512) if 0 <= r0 + d * 32 < M}
            print("   first bad", r0, c0, "got addend", got_add, "candidates by slab offset", cands, flush=True)
print("bad runs:", bad_runs)
