"""Drop-in replacement for the CUDA package the reference imports at
/root/reference/gaussian_renderer/__init__.py:6 —

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

Putting this repository on PYTHONPATH makes that line resolve here; the implementation is the
gfx950 HIP rasterizer in gaussianavatar_amd (C ABI: include/gsr.h)."""
from gaussianavatar_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
