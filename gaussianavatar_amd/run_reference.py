"""Runs one of the reference's scripts (train.py, eval.py, render_novel_pose.py) UNCHANGED on this
repository:

    python -m gaussianavatar_amd.run_reference /path/to/GaussianAvatar/train.py -s <data> -m <out> ...

Python puts a script's own directory first on sys.path, so `python train.py` inside the reference
tree would pick up the reference's `model/`, `utils/`, `scene/`, `gaussian_renderer/` packages (and
then fail on `diff_gaussian_rasterization`, CUDA-only). This launcher executes the same file with
`dropin/` and the repository root in front instead: `model.avatar_model.AvatarModel`,
`gaussian_renderer.render_batch`, `utils.loss_utils.ssim`, ... resolve to the MI355X implementations
(dropin/README.md lists the map), `diff_gaussian_rasterization` to the HIP rasterizer. The script's
bytes, its command line and its working-directory conventions (`project_path = os.getcwd()`) stay as
they are.
"""
from __future__ import annotations

import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "dropin")
ALIASES = ("model", "scene", "utils", "arguments", "gaussian_renderer")


def install_paths() -> None:
    """dropin/ and the repository root go to the front of sys.path; already-imported modules that
    would shadow the aliases (another `utils`, the reference's `model`) are dropped."""
    for p in (ROOT, DROPIN):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for name in list(sys.modules):
        top = name.split(".")[0]
        if top in ALIASES:
            f = getattr(sys.modules[name], "__file__", None) or ""
            if not os.path.abspath(f).startswith(DROPIN):
                del sys.modules[name]


def run(script: str, argv=(), run_name: str = "__main__") -> dict:
    """Executes `script` (a path, any extension) as `run_name` with sys.argv = [script, *argv].
    Returns the script's globals."""
    install_paths()
    old_argv = sys.argv
    sys.argv = [script] + [str(a) for a in argv]
    try:
        with open(script, "rb") as f:
            code = compile(f.read(), script, "exec")
        glb = {"__name__": run_name, "__file__": script, "__builtins__": __builtins__}
        exec(code, glb)
        return glb
    finally:
        sys.argv = old_argv


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    run(sys.argv[1], sys.argv[2:])
