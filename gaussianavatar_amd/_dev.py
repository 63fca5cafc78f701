"""Development switches of the Python layer — the only place the package reads the environment for behaviour.

ONE variable, `GA_DEV="key=value,key=value"`, read once at import; without it every field has its product value.
(The HIP libraries read no environment at all; what used to be GANET_MFMA / GANET_BWD_SWEEP / GSR_ABLATE is gone.)

    lib_dir=<path>      load libgsr / libgalbs / libganet from <path> instead of gaussianavatar_amd/_lib
                        (A/B runs of a variant build, tools/build_variant.sh)
    wgrad_stream=0      decoder backward: the weight-gradient launches on the main stream instead of the side stream
    unet_wgrad_stream=0 stage 2: the pose encoder's weight gradients on the main stream (ganet_unet_bwd without a side stream)
    encoder_stream=0    stage 2: the pose encoder on the main stream (default: a side stream of its own, beside the geometry net and the
                        body model forward, and — autograd runs a node's backward on its forward's stream — beside their backward)
    row_sweep=0         decoder: every launch sweeps the rows first-to-last (no alternation)
    native_decoder=0    decoder: the per-layer launch sequence from Python instead of one native call each way
    one_pass_backward=0 decoder: separate weight- / data-gradient kernels for the hidden layers
    native_unet=0       stage 2: the pose encoder as im2col + vendor GEMM + torch BatchNorm / element-wise kernels instead of
                        the hand-written kernels of csrc/ganet_unet.hip
"""
from __future__ import annotations

import os
from dataclasses import dataclass, fields


@dataclass
class DevKnobs:
    lib_dir: str = ""
    wgrad_stream: bool = True
    unet_wgrad_stream: bool = True
    encoder_stream: bool = True
    row_sweep: bool = True
    native_decoder: bool = True
    one_pass_backward: bool = True
    native_unet: bool = True


def _parse(text: str) -> DevKnobs:
    k = DevKnobs()
    names = {f.name: f.type for f in fields(DevKnobs)}
    for item in filter(None, (p.strip() for p in text.split(","))):
        key, _, val = item.partition("=")
        if key not in names:
            raise ValueError(f"GA_DEV: unknown key {key!r} (known: {', '.join(names)})")
        setattr(k, key, val if key == "lib_dir" else val not in ("0", "false", "off", ""))
    return k


knobs = _parse(os.environ.get("GA_DEV", ""))
