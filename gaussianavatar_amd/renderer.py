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
