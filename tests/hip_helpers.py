"""Helpers shared by the GPU parity tests: run the HIP rasterizer on a tests.scenes scene."""
from __future__ import annotations

import numpy as np
import torch


def settings_from_scene(sc, device="cuda", debug=False):
    from gaussianavatar_amd.rasterizer import GaussianRasterizationSettings
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=device)
    return GaussianRasterizationSettings(
        image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"],
        bg=t(sc["bg"]), scale_modifier=sc.get("scale_modifier", 1.0), viewmatrix=t(sc["viewmatrix"]),
        projmatrix=t(sc["projmatrix"]), sh_degree=0, campos=t(sc["campos"]), prefiltered=False,
        debug=debug)


def scene_tensors(sc, device="cuda", requires_grad=False):
    out = {}
    for k in ("means3D", "colors", "opacities", "scales", "rotations"):
        v = torch.tensor(sc[k], dtype=torch.float32, device=device)
        if k == "opacities":
            v = v.reshape(-1, 1)
        out[k] = v.requires_grad_(requires_grad)
    return out


def hip_forward_state(sc, max_pairs=None):
    """Forward through the C ABI; returns numpy copies of the outputs and the internal state."""
    from gaussianavatar_amd.rasterizer import rasterize_with_state
    rs = settings_from_scene(sc)
    t = scene_tensors(sc)
    color, radii, views, status = rasterize_with_state(
        rs, t["means3D"], t["colors"], t["opacities"], t["scales"], t["rotations"], max_pairs=max_pairs)
    st = {k: v.cpu().numpy() for k, v in views.items() if k not in ("grad_acc",)}
    st["color"] = color.cpu().numpy()
    st["radii"] = radii.cpu().numpy()
    st["status"] = status
    return st


def hip_tile_lists(st):
    off = st["tile_offset"].astype(np.int64)
    pl = st["point_list"]
    return [pl[off[t]:off[t + 1]] for t in range(len(off) - 1)]


def hip_forward_backward(sc, grad_out):
    """Autograd round trip through GaussianRasterizer; returns (color, grads dict) as numpy."""
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    rs = settings_from_scene(sc)
    t = scene_tensors(sc, requires_grad=True)
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    color, radii = GaussianRasterizer(rs)(
        means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=None,
        colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    g = torch.tensor(grad_out, dtype=torch.float32, device=color.device)
    color.backward(g)
    grads = dict(dmeans3D=t["means3D"].grad, dmeans2D=means2D.grad, dcolors=t["colors"].grad,
                 dopacity=t["opacities"].grad.reshape(-1), dscales=t["scales"].grad,
                 drots=t["rotations"].grad)
    return color.detach().cpu().numpy(), radii.cpu().numpy(), {k: v.cpu().numpy() for k, v in grads.items()}
