"""`render_batch` — mirror of /root/reference/gaussian_renderer/__init__.py:8-50.

Same signature and return value (image [3,H,W]). Differences, all invisible to the caller:
  * FovX/FovY/height/width may be python scalars (preferred) or 0-d tensors. The reference
    feeds 0-d CUDA tensors to math.tan()/int(), which costs four implicit host syncs per frame
    (SURVEY.md Appendix B); python scalars cost none. Tensors are still accepted.
  * no `screenspace_points` tensor is allocated when nobody asked for its gradient.
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _scalar(v):
    return v.item() if torch.is_tensor(v) else v


def render_batch(points, shs, colors_precomp, rotations, scales, opacity, FovX, FovY, height, width,
                 bg_color, world_view_transform, full_proj_transform, active_sh_degree, camera_center,
                 screenspace_points=None):
    tanfovx = math.tan(_scalar(FovX) * 0.5)
    tanfovy = math.tan(_scalar(FovY) * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(_scalar(height)),
        image_width=int(_scalar(width)),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=1.0,
        viewmatrix=world_view_transform,
        projmatrix=full_proj_transform,
        sh_degree=active_sh_degree,
        campos=camera_center,
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    rendered_image, _radii = rasterizer(
        means3D=points,
        means2D=screenspace_points,
        shs=shs,
        colors_precomp=colors_precomp,
        opacities=opacity,
        scales=scales,
        rotations=rotations,
        cov3D_precomp=None)
    return rendered_image


def render_frames(points, colors_precomp, rotations, scales, opacity, FovX, FovY, height, width, bg_color,
                  world_view_transform, full_proj_transform, camera_center):
    """All frames of a batch in one call: points [B,N,3], colors_precomp / scales [B,N,3] (expanded
    views are read once), rotations [N,4], opacity [N,1], per-frame camera lists/tensors as the
    reference's batch dict carries them (/root/reference/model/avatar_model.py:333-339).

    Frames that share image size and FoV (every dataset of the reference: one static camera) go
    through ONE launch of each rasterizer kernel (gsr_forward_batch); otherwise this falls back
    to render_batch per frame. Returns [B,3,H,W]."""
    from .rasterizer import rasterize_gaussians_batch
    B = points.shape[0]
    fx = [float(_scalar(FovX[b])) for b in range(B)]
    fy = [float(_scalar(FovY[b])) for b in range(B)]
    hs = [int(_scalar(height[b])) for b in range(B)]
    ws = [int(_scalar(width[b])) for b in range(B)]
    uniform = all(v == fx[0] for v in fx) and all(v == fy[0] for v in fy) and \
        all(v == hs[0] for v in hs) and all(v == ws[0] for v in ws)
    if not uniform:
        return torch.stack([render_batch(points[b], None, colors_precomp[b], rotations, scales[b], opacity,
                                         fx[b], fy[b], hs[b], ws[b], bg_color, world_view_transform[b],
                                         full_proj_transform[b], 0, camera_center[b]) for b in range(B)], dim=0)
    settings = GaussianRasterizationSettings(
        image_height=hs[0], image_width=ws[0], tanfovx=math.tan(fx[0] * 0.5), tanfovy=math.tan(fy[0] * 0.5),
        bg=bg_color, scale_modifier=1.0, viewmatrix=world_view_transform, projmatrix=full_proj_transform,
        sh_degree=0, campos=camera_center, prefiltered=False, debug=False)
    image, _radii = rasterize_gaussians_batch(points, colors_precomp, opacity, scales, rotations, settings)
    return image
