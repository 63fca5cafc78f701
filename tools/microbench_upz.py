"""development: stand-alone timings of the map-path kernels (csrc/ganet_upz.hip, ganet_layer_fwd.hip ADD) at the headline
size (S = 512, R = 128, one frame), inputs rotated through > 256 MB so that they come from HBM."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native
from tests.test_decoder_map_gpu import _grid_setup, _ptr, _stream

lib = _native.ganet()
b, feat, S = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 128, 512
net, uv, mats, taps, grid = _grid_setup(b, feat, S)
M = b * S * S
NB = 3
Gs = [torch.randn(M, 128, device="cuda") for _ in range(NB)]
Zs = [torch.randn(M, 128, device="cuda") for _ in range(NB)]
Os = [torch.empty(M, 128, device="cuda") for _ in range(NB)]
P = torch.randn(b * feat * feat, 256, device="cuda")
fmap = torch.randn(b * feat * feat, 64, device="cuda")
Wf = torch.randn(256, 64, device="cuda")
WfT = torch.randn(64, 256, device="cuda")
dP = torch.empty(b * feat * feat, 256, device="cuda")
dfeat = torch.empty(b * feat * feat, 64, device="cuda")
coef = torch.randn(3, 128, device="cuda")
Wuv = torch.randn(128, 2, device="cuda")
bias = torch.randn(128, device="cuda")
sc = torch.rand(128, device="cuda") + 0.5
sh = torch.randn(128, device="cuda") * 0.3
W = torch.randn(128, 128, device="cuda") * 0.1
cp = torch.empty(lib.ganet_mlp_stats_floats(128), device="cuda")
nparts = lib.ganet_dz_upsample_t_parts(ctypes.byref(grid))
part = torch.empty(nparts, 384, device="cuda")
st = _stream()


def timeit(name, fn, bytes_=None, reps=21):
    fn(0); torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record()
    for k in range(reps):
        fn(k)
        evs[k + 1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(evs, evs[1:]))
    med = ts[len(ts) // 2]
    extra = f"  {bytes_ / med / 1e6:.2f} TB/s" if bytes_ else ""
    print(f"{name:28s} median {med:8.1f} us  min {ts[0]:8.1f}{extra}", flush=True)


chk = _native.ganet_check
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
if ONLY == "dz":
    timeit("dz_upsample_t", lambda k: chk(lib.ganet_dz_upsample_t(ctypes.byref(grid), _ptr(Gs[k % NB]), _ptr(Zs[k % NB]), _ptr(coef), _ptr(dP[:, 128:]), 256, _ptr(part), st)), 8.0 * M * 128)
    sys.exit(0)
big = [torch.empty(M, 128, device="cuda") for _ in range(NB)]
timeit("torch zero_ (134 MB write)", lambda k: big[k % NB].zero_(), 4.0 * M * 128)
timeit("torch sum (134 MB read)", lambda k: Gs[k % NB].sum(), 4.0 * M * 128)
timeit("torch copy_ (134 r + 134 w)", lambda k: big[k % NB].copy_(Gs[k % NB]), 8.0 * M * 128)
timeit("torch add (268 r + 134 w)", lambda k: torch.add(Gs[k % NB], Zs[k % NB], out=big[k % NB]), 12.0 * M * 128)
timeit("rowgemm P (N256 K64)", lambda k: chk(lib.ganet_rowgemm(b * feat * feat, 256, 64, _ptr(fmap), 64, _ptr(Wf), 64, _ptr(P), 256, 0, st)))
timeit("rowgemm dfeat (N64 K256)", lambda k: chk(lib.ganet_rowgemm(b * feat * feat, 64, 256, _ptr(dP), 256, _ptr(WfT), 256, _ptr(dfeat), 64, 0, st)))
timeit("upsample_z_fwd", lambda k: chk(lib.ganet_upsample_z_fwd(ctypes.byref(grid), _ptr(P), 256, _ptr(Wuv), _ptr(bias), _ptr(sh), _ptr(Os[k % NB]), _ptr(cp), st)), 4.0 * M * 128)
timeit("mlp_fwd_add (skip layer)", lambda k: chk(lib.ganet_mlp_fwd_add(ctypes.byref(grid), _ptr(Gs[k % NB]), _ptr(sc), _ptr(sh), _ptr(W), _ptr(bias), _ptr(P[:, 128:]), 256, _ptr(Wuv), _ptr(Os[k % NB]), _ptr(cp), _ptr(sh), 0, st)), 8.0 * M * 128)
timeit("mlp_fwd (hidden layer)", lambda k: chk(lib.ganet_mlp_fwd(M, 128, 0, 128, None, 0, _ptr(Gs[k % NB]), 128, _ptr(sc), _ptr(sh), _ptr(W), _ptr(bias), _ptr(Os[k % NB]), 128, _ptr(cp), _ptr(sh), 0, st)), 8.0 * M * 128)
timeit("dz_upsample_t", lambda k: chk(lib.ganet_dz_upsample_t(ctypes.byref(grid), _ptr(Gs[k % NB]), _ptr(Zs[k % NB]), _ptr(coef), _ptr(dP[:, 128:]), 256, _ptr(part), st)), 8.0 * M * 128)
ws = torch.empty(lib.ganet_wgrad_act_workspace(b * feat * feat, 128, 64), dtype=torch.uint8, device="cuda")
dW = torch.empty(128, 64, device="cuda"); db = torch.empty(128, device="cuda")
timeit("wgrad_act dWf (M16384)", lambda k: chk(lib.ganet_wgrad_act(b * feat * feat, 128, 64, _ptr(dP), 256, None, 0, None, _ptr(fmap), 64, None, None, _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), 0, st)))
