#!/usr/bin/env python
"""Hidden-layer forward at M = 262,144: one launch (layer_fwd_spec_kernel<1>) and the three-branch launch (<3>), HIP-event
averages. With a -DGANET_LFWD_ABLATE=<bits> build (tools/lfwd_ablate.sh) the same numbers with parts of the round removed:
which part of the round bounds the kernel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
dev = torch.device("cuda"); M = 262144
torch.manual_seed(0)
x = torch.randn(3, M, 128, device=dev); W = torch.randn(3, 128, 128, device=dev) * 0.1; b = torch.randn(3, 128, device=dev)
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
z = torch.empty(3, M, 128, device=dev); part = torch.zeros(3, 256, 256, device=dev)
st = fused._stream(dev); P = fused._ptr
lib.ganet_dev_layer_fwd3.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 8
def one():
    _native.ganet_check(lib.ganet_mlp_fwd(M, 128, 0, 128, None, 0, P(x), 128, P(sc), P(sh), P(W), P(b), P(z), 128, P(part), None, 1, st))
def three():
    _native.ganet_check(lib.ganet_dev_layer_fwd3(M, P(x), P(sc), P(sh), P(W), P(b), P(z), P(part), st))
out = []
for name, fn in (("one", one), ("three", three)):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): fn()
    e1.record(); torch.cuda.synchronize()
    out.append("%s %.1f us" % (name, e0.elapsed_time(e1) / 40 * 1e3))
print(os.environ.get("GA_DEV", "product"), " | ".join(out))
