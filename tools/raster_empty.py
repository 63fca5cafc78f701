#!/usr/bin/env python
"""Rasterizer kernel times for a scene whose Gaussians are all behind the camera (every tile empty) next to a normal
one: what the launch costs before it blends anything (2 frames, 200k Gaussians, 1024^2)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import rasterizer as R
dev = torch.device("cuda"); B, P, S = 2, 200_000, 1024
torch.manual_seed(0)
def run(zoff, label):
    pts = torch.randn(B, P, 3, device=dev) * 0.3 + torch.tensor([0., 0., zoff], device=dev)
    pts.requires_grad_()
    col = torch.rand(P, 3, device=dev); opa = torch.ones(P, 1, device=dev)
    sca = torch.full((P, 3), 0.004, device=dev); rot = torch.zeros(P, 4, device=dev); rot[:, 0] = 1
    view = torch.eye(4, device=dev)[None].expand(B, -1, -1).contiguous()
    fov = 0.6; t = math.tan(fov / 2); n, f = 0.01, 100.0
    Pm = torch.zeros(4, 4, device=dev); Pm[0, 0] = 1 / t; Pm[1, 1] = 1 / t; Pm[3, 2] = 1.0; Pm[2, 2] = f / (f - n); Pm[2, 3] = -(f * n) / (f - n)
    proj = (view[0] @ Pm.t())[None].expand(B, -1, -1).contiguous()
    st = R.GaussianRasterizationSettings(image_height=S, image_width=S, tanfovx=t, tanfovy=t, bg=torch.ones(3, device=dev),
                                         scale_modifier=1.0, viewmatrix=view, projmatrix=proj, sh_degree=0,
                                         campos=torch.zeros(B, 3, device=dev), prefiltered=False, debug=False)
    for k in range(12):
        if k == 4: R.profile_enable(True); R.profile_read(True)
        img, _ = R.rasterize_gaussians_batch(pts, col, opa, sca, rot, st)
        img.sum().backward(); pts.grad = None
    torch.cuda.synchronize()
    r = R.profile_read(True); R.profile_enable(False)
    print(label, "pairs/frame", R.last_status()[0] if R.last_status() else None, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in r.items() if v[1]})
run(-3.0, "all behind the camera:")
run(+3.0, "in front:")
