"""development: the hidden-layer forward (layer_fwd_spec_kernel through fused._mlp_fwd) at M = 262,144, NaN-prefilled output,
against float64 — first call of a fresh process and repeats; describes any mismatch (unwritten / wrong; slabs, rows, columns)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from gaussianavatar_amd import _native, fused

lib = _native.ganet()
M, K2, N = 262144, 128, 128
torch.manual_seed(M % 89)
dev = "cuda"
x2 = torch.randn(M, K2, device=dev) * 2 + torch.linspace(-4, 4, K2, device=dev)
sc = torch.empty(K2, device=dev).uniform_(0.3, 2.0)
sh = torch.empty(K2, device=dev).uniform_(-25, 25)
W = torch.randn(N, K2, device=dev) * 0.1
b = torch.randn(N, device=dev)
ref = F.softplus(x2.double() * sc.double() + sh.double()) @ W.double().t() + b.double()
tol = 2e-5 * float(ref.abs().max()) + 1e-5
bad_runs = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
orig_empty = torch.empty
for it in range(reps):
    for order in (0, 2):
        part = torch.zeros(lib.ganet_mlp_stats_floats(N), device=dev)
        # fused._mlp_fwd allocates its output with torch.empty: make that NaN-filled to see unwritten rows
        def nan_empty(*a, **kw):
            t = orig_empty(*a, **kw)
            if t.is_floating_point() and t.is_cuda:
                t.fill_(float("nan"))
            return t
        torch.empty = nan_empty
        try:
            z = fused._mlp_fwd(lib, M, N, None, x2, sc, sh, W, b, part, dev, order)
        finally:
            torch.empty = orig_empty
        err = (z.double() - ref).abs()
        badm = ~(err <= tol)
        nb = int(badm.sum())
        if nb:
            bad_runs += 1
            idx = torch.nonzero(badm)
            rows, cols = idx[:, 0], idx[:, 1]
            slabs = torch.unique(rows // 32)
            print(f"call {it} order {order}: {nb} bad elements, nan {int(torch.isnan(z).sum())}, {slabs.numel()} slabs (first {slabs[:8].tolist()}), "
                  f"rows-in-slab {torch.unique(rows % 32)[:32].tolist()}, n_cols {torch.unique(cols).numel()} cols {torch.unique(cols)[:8].tolist()}, "
                  f"max err {float(err[~torch.isnan(err)].max()):.3f}", flush=True)
            r0, c0 = int(rows[0]), int(cols[0])
            print("   first bad", r0, c0, "got", float(z[r0, c0]), "want", float(ref[r0, c0]),
                  "other slabs same (row-in-slab, col):", [round(float(ref[(r0 % 32) + 32 * s, c0]), 4) for s in (r0 // 32 - 1, r0 // 32 + 1, r0 // 32 + 256, r0 // 32 - 256) if 0 <= s < M // 32], flush=True)
print("bad runs:", bad_runs, "of", 2 * reps)
