"""Chained decoder-backward layers (wgrad_i, bwd_data_i, wgrad_{i-1}, ...) with the common row front
swept in alternating directions vs the plain order: does a kernel find the rows the previous one touched
last in the Infinity Cache? (dev tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
dev = torch.device("cuda")
M = 262144
zs = [torch.randn(M, 128, device=dev) for _ in range(5)]
Gs = [torch.randn(M, 128, device=dev) for _ in range(2)]
sc = torch.rand(128, device=dev) + 0.5
sh = torch.randn(128, device=dev)
W = torch.randn(128, 128, device=dev) * 0.1
coef = torch.randn(3, 128, device=dev)
dW = torch.empty(128, 128, device=dev); db = torch.empty(128, device=dev)
nb = lib.ganet_wgrad_act_workspace(M, 128, 128)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
part = torch.zeros(lib.ganet_mlp_bwd_data_parts() * 256, device=dev)
st = fused._stream(dev)
P = fused._ptr
def chain(mode):
    for i in (4, 3, 2, 1):
        g_in, g_out = Gs[i % 2], Gs[(i + 1) % 2]
        lib.ganet_wgrad_act(M, 128, 128, P(g_in), 128, P(zs[i]), 128, P(coef), P(zs[i - 1]), 128, P(sc), P(sh),
                            P(dW), P(db), P(ws), nb, {0: 0, 1: 1, 2: 1}[mode], st)
        lib.ganet_mlp_bwd_data(M, 128, P(g_in), 128, P(zs[i]), 128, P(coef), P(W), 128, P(g_out), 128, 0,
                               P(zs[i - 1]), 128, P(sc), P(sh), P(part), {0: 0, 1: 1, 2: 2}[mode], st)
names = {0: "contiguous wgrad ranges, ascending", 1: "common front, all ascending", 2: "common front, alternating"}
for rep in range(2):
    for mode in (0, 1, 2):
        for _ in range(3): chain(mode)
        torch.cuda.synchronize()
        fused.profile_enable(["wgrad_act", "mlp_bwd_data"]); fused.profile_read(True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): chain(mode)
        e.record(); torch.cuda.synchronize()
        pr = fused.profile_read(True); fused.profile_enable(False)
        print("%-38s %.1f us per layer (wgrad %.1f, bwd_data %.1f)" % (
            names[mode], s.elapsed_time(e) / 40 * 1e3, pr["wgrad_act"][0] / pr["wgrad_act"][1] * 1e3,
            pr["mlp_bwd_data"][0] / pr["mlp_bwd_data"][1] * 1e3))
