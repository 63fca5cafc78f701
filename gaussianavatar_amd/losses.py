"""Image losses the reference's train loop applies to the rendered batch
(/root/reference/train.py:74-75): L1 (`l1_loss_w`, utils/loss_utils.py:7-8) and SSIM with an
11x11 Gaussian window, sigma 1.5, zero padding (utils/loss_utils.py:10-53).

The window is separable (outer product of a 1-D Gaussian), so the five grouped 11x11
convolutions of the reference are evaluated as 1-D row/column passes over a stacked
[mu1, mu2, x1^2, x2^2, x1*x2] tensor: same value up to float rounding, ~5x fewer MACs and one
conv launch pair instead of five.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# The training loop evaluates l1_loss_w(image, gt) and ssim(image, gt) back to back on the same pair
# (/root/reference/train.py:74-75). On a HIP device both come out of ONE pass over the images
# (fused.ssim_l1_mean) and share one backward pass: whichever is asked for first computes the pair and
# parks the other value here for the matching call (same tensors, same in-place versions).
_pair = {}


def _fusable(img1, img2, window_size=11, size_average=True):
    return (img1.is_cuda and size_average and window_size == 11 and img1.dtype == torch.float32
            and img1.shape == img2.shape and img1.dim() in (3, 4) and not img2.requires_grad)


def _paired(img1, img2, want: str):
    # grad mode is part of the key: a value parked under no_grad (logging) has no graph and must not
    # answer a later call that wants gradients
    key = (id(img1), img1._version, id(img2), img2._version, torch.is_grad_enabled(), img1.requires_grad)
    ent = _pair.get("entry")
    if ent is not None and ent["key"] == key and ent["a"]() is img1 and ent["b"]() is img2 and want in ent["left"]:
        out = ent["left"].pop(want)
        if not ent["left"]:
            _pair.clear()
        return out
    import weakref
    from . import fused
    s, l = fused.ssim_l1_mean(img1, img2.to(img1.dtype))
    vals = {"ssim": s, "l1": l}
    out = vals.pop(want)
    _pair["entry"] = {"key": key, "a": weakref.ref(img1), "b": weakref.ref(img2), "left": vals}
    return out


def l1_loss_w(network_output, gt):
    if _fusable(network_output, gt):
        return _paired(network_output, gt, "l1")
    return torch.abs(network_output - gt).mean()


def _gauss_1d(window_size: int, sigma: float, device, dtype):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2))
                      for x in range(window_size)], dtype=torch.float32)
    return (g / g.sum()).to(device=device, dtype=dtype)


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    """img [..., C, H, W] (3-D or 4-D like the reference accepts)."""
    if _fusable(img1, img2, window_size, size_average):
        return _paired(img1, img2, "ssim")       # one streamed HIP pass forward, one backward
    squeeze = img1.dim() == 3
    if squeeze:
        img1, img2 = img1[None], img2[None]
    B, C, H, W = img1.shape
    g = _gauss_1d(window_size, 1.5, img1.device, img1.dtype)
    pad = window_size // 2
    stack = torch.cat([img1, img2, img1 * img1, img2 * img2, img1 * img2], dim=1)    # [B,5C,H,W]
    n = stack.shape[1]
    kx = g.view(1, 1, 1, -1).expand(n, 1, 1, window_size)
    ky = g.view(1, 1, -1, 1).expand(n, 1, window_size, 1)
    f = F.conv2d(F.conv2d(stack, kx, padding=(0, pad), groups=n), ky, padding=(pad, 0), groups=n)
    mu1, mu2, s11, s22, s12 = f.split(C, dim=1)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    sigma1_sq, sigma2_sq, sigma12 = s11 - mu1_sq, s22 - mu2_sq, s12 - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)


def adjust_loss_weights(init_weight, current_epoch, mode="decay", start=400, every=20):
    """Regulariser schedule of the training loop (train.py:59 via utils/general_utils.py:261-280):
    constant until `start` ('rise' starts at 1e-6 of the weight), then x0.85 ('decay') or x1.05
    ('rise') per `every` epochs; every=0 keeps it constant."""
    if mode not in ("decay", "rise"):
        raise ValueError("mode must be 'decay' or 'rise'")
    if current_epoch < start:
        return init_weight * 1e-6 if mode == "rise" else init_weight
    if every == 0:
        return init_weight
    steps = (current_epoch - start) // every
    return init_weight * ((1.05 if mode == "rise" else 0.85) ** steps)


def weighted_sum(terms, weights, bias: float = 0.0):
    """bias + sum_i weights[i] * terms[i] over zero-dimensional loss terms — the way the training loop
    composes its objective (train.py:70-82: `scale_loss + offset_loss + Ll1 + ssim_loss + geo_loss` with
    python-number weights, `1 - ssim` = bias 1 and weight -1). On a HIP device this is one launch each
    way instead of one zero-dimensional kernel per arithmetic operator."""
    terms = list(terms)
    if terms and all(torch.is_tensor(t) and t.is_cuda and t.numel() == 1 for t in terms) and len(terms) <= 8:
        from . import fused
        return fused.weighted_sum(terms, weights, bias)
    out = bias
    for t, w in zip(terms, weights):
        out = out + w * t
    return out
