"""Builds the gfx950 shared libraries in-tree with hipcc (no torch headers involved).

    python -m gaussianavatar_amd.build [--force]

Outputs (git-ignored, but they travel to the GPU box with the repo snapshot):
    gaussianavatar_amd/_lib/libgsr_hip.so     rasterizer   (include/gsr.h)
    gaussianavatar_amd/_lib/libgalbs_hip.so   LBS kernels  (include/galbs.h)
    gaussianavatar_amd/_lib/libganet_hip.so   fused net/loss kernels (include/ganet.h)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib")
OBJ = os.path.join(OUT, "obj")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
          "-I" + CSRC, "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]

# (source, extra flags). gsr_preprocess.hip decides the integer outputs (radii, tile rects):
# no FMA contraction there, so that it follows the operation order shared with the oracle.
LIBS = {
    "libgsr_hip.so": [
        ("gsr_api.hip", []),
        ("gsr_preprocess.hip", ["-ffp-contract=off"]),
        ("gsr_binning.hip", []),
        ("gsr_render.hip", []),
        ("gsr_sh.hip", []),
    ],
    "libgalbs_hip.so": [
        ("galbs.hip", []),
    ],
    "libganet_hip.so": [
        ("ganet_bn.hip", []),
        ("ganet_wgrad.hip", []),
        ("ganet_ssim.hip", []),
        ("ganet_mlp.hip", []),
        ("ganet_mlp_bwd.hip", []),
        ("ganet_mlp_split.hip", []),
        ("ganet_wgrad_split.hip", []),
        ("ganet_layer_bwd.hip", []),
        ("ganet_layer_fwd.hip", []),
        ("ganet_decoder.hip", []),
        ("ganet_pack.hip", []),
        ("ganet_upsample.hip", []),
        ("ganet_unet.hip", []),
        ("ganet_conv.hip", []),
        ("ganet_optim.hip", []),
    ],
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> dict:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    built = {}
    for lib, sources in LIBS.items():
        sources = [(s, fl) for s, fl in sources if os.path.exists(os.path.join(CSRC, s))]
        if not sources:
            continue
        objs = []
        for src, flags in sources:
            sp = os.path.join(CSRC, src)
            op = os.path.join(OBJ, src.replace(".hip", ".o"))
            if force or not _newer(op, [sp] + headers):
                cmd = [hipcc] + COMMON + flags + ["-c", sp, "-o", op]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
            objs.append(op)
        target = os.path.join(OUT, lib)
        if force or not _newer(target, objs):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", target] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        built[lib] = target
    return built


if __name__ == "__main__":
    out = build(force="--force" in sys.argv, verbose=True)
    for k, v in out.items():
        print(k, "->", v)
