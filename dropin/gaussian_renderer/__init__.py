"""`gaussian_renderer` as the reference's model imports it (/root/reference/model/avatar_model.py:14).
The reference's own 50-line shim (/root/reference/gaussian_renderer/__init__.py) also runs unchanged on the
top-level `diff_gaussian_rasterization` package of this repository; this alias is the sync-free variant."""
from gaussianavatar_amd.renderer import render_batch, render_frames  # noqa: F401
