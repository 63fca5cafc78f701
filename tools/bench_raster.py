#!/usr/bin/env python
"""Rasterizer-only timing on the headline frame (200k avatar Gaussians, 1024^2): per-kernel HIP
event averages over repeated forward+backward calls. GSR_ABLATE=<bits> attributes time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import rasterizer
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
from gaussianavatar_amd.renderer import render_batch
from gaussianavatar_amd.lbs import skin

torch.manual_seed(0)
N = int(os.environ.get("POINTS", 200000)); size = int(os.environ.get("SIZE", 1024))
mp, npar, op = default_params(batch_size=1, num_points=N, image_width=size, image_height=size)
m = AvatarModel(mp, npar, op, train=True)
batch = collate_frames([m.train_dataset[0]], "cuda")
with torch.no_grad():
    live = m._body(m.pose.weight[:1], m.transl.weight[:1], None)
    _off, _scl, point_res, scales, colors = m._decode(1, None, 7, True)
    pts = skin(m.query_points[:1], point_res, m.query_lbs[0], live.cano2live)[0].contiguous()
    scales, colors = scales[0].contiguous(), colors[0].contiguous()
pts.requires_grad_(True); scales.requires_grad_(True); colors.requires_grad_(True)
g = torch.randn(3, size, size, device="cuda")
def it():
    img = render_batch(pts, None, colors, m.fix_rotation, scales, m.fix_opacity, batch["FovX"][0], batch["FovY"][0],
                       size, size, m.background, batch["world_view_transform"][0], batch["full_proj_transform"][0], 0,
                       batch["camera_center"][0])
    img.backward(g)
    pts.grad = scales.grad = colors.grad = None
for _ in range(5): it()
rasterizer.check_overflow(True); rasterizer.pair_statistics(reset=True)
rasterizer.profile_enable(True); rasterizer.profile_read(True)
reps = int(os.environ.get("REPS", 30))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): it()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e6
prof = rasterizer.profile_read(True)
n, pairs = rasterizer.pair_statistics(True)
print(f"ABLATE={os.environ.get('GSR_ABLATE','0')} pairs/frame={pairs:.0f} wall fwd+bwd={wall:.0f}us  " +
      "  ".join(f"{k}={ms/c*1e3:.1f}" for k, (ms, c) in prof.items() if c))

# forward-only renders (torch.no_grad(): the reference's eval.py / render_novel_pose.py): no segment records, small workspace;
# two frames per launch as in the training iteration, per-kernel HIP events
from gaussianavatar_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians_batch
import math
B = 2
bt = collate_frames([m.train_dataset[i] for i in range(B)], "cuda")
rs = GaussianRasterizationSettings(size, size, math.tan(float(bt["FovX"][0]) * 0.5), math.tan(float(bt["FovY"][0]) * 0.5), m.background,
                                   1.0, bt["world_view_transform"][:B], bt["full_proj_transform"][:B], 0, bt["camera_center"][0], False, False)
pts2 = torch.stack([pts.detach(), pts.detach() + 0.003]).contiguous()
for grad in (False, True):
    def it2():
        p_ = pts2.clone().requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            return rasterize_gaussians_batch(p_, colors.detach(), m.fix_opacity, scales.detach(), m.fix_rotation, rs)[0]
    for _ in range(5): it2()
    rasterizer.check_overflow(True); rasterizer.profile_read(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): it2()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e6
    prof = rasterizer.profile_read(True)
    print(f"2 frames per launch, forward only, {'recording (requires_grad)' if grad else 'no_grad (gsr_forward_eval_batch)'}: wall {wall:.0f} us  " +
          "  ".join(f"{k}={ms/c*1e3:.1f}" for k, (ms, c) in prof.items() if c))
