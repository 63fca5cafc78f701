"""Autograd wrappers of the fused feature-net / loss kernels (include/ganet.h).

Used by network.ShapeDecoder and losses.ssim whenever their tensors live on a HIP device; on
the CPU the same modules run the plain torch formulation (that is what the golden-vector tests
exercise), so the two formulations are checked against each other on the GPU box
(tests/test_fused_gpu.py).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from . import _native


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def wgrad_supported(N: int, K: int) -> bool:
    return N <= 128 and K <= 224


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b with x [M,K], W [N,K] (the 1x1-conv weight squeezed), b [N]. Forward and
    the input gradient are vendor GEMMs (fast for these shapes); the weight/bias gradient — a
    reduction over M = 262,144 rows — is the MFMA split kernel ganet_linear_wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = g @ weight if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            M, K = x.shape
            N = weight.shape[0]
            if wgrad_supported(N, K):
                lib = _native.ganet()
                xc = x if x.stride(1) == 1 else x.contiguous()
                dW = torch.empty(N, K, dtype=torch.float32, device=x.device)
                db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
                nbytes = lib.ganet_linear_wgrad_workspace(M, N, K)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                _native.ganet_check(lib.ganet_linear_wgrad(
                    M, N, K, _ptr(g), g.stride(0), _ptr(xc), xc.stride(0), _ptr(dW), _ptr(db), _ptr(ws),
                    nbytes, _stream(x.device)))
            else:
                dW = g.t() @ x
                db = g.sum(0) if ctx.has_bias else None
        return dx, dW, db


def linear(x, weight, bias):
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        return _LinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def bn_supported(C: int) -> bool:
    return C % 4 == 0 and C <= 256 and 256 % (C // 4) == 0


class _BnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, running_mean, running_var, momentum, num_batches_tracked):
        lib = _native.ganet()
        M, C = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        nbytes = lib.ganet_bn_workspace(M, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _native.ganet_check(lib.ganet_bn_act_fwd(M, C, _ptr(x), _ptr(gamma), _ptr(beta), float(eps), int(act),
                                                 _ptr(y), _ptr(mean), _ptr(rstd), _ptr(running_mean),
                                                 _ptr(running_var), float(momentum), _ptr(num_batches_tracked),
                                                 _ptr(ws), nbytes, _stream(x.device)))
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.act = int(act)
        ctx.mark_non_differentiable(mean, rstd)
        return y, mean, rstd

    @staticmethod
    def backward(ctx, dy, _dmean, _drstd):
        lib = _native.ganet()
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        M, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        nbytes = lib.ganet_bn_workspace(M, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _native.ganet_check(lib.ganet_bn_act_bwd(M, C, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd),
                                                 ctx.act, _ptr(dy), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                                 _ptr(ws), nbytes, _stream(x.device)))
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batchnorm_act(x, bn: torch.nn.BatchNorm1d, act: str = "softplus"):
    """act(bn(x)) for x [M,C] with training-mode (batch) statistics, updating the module's running
    statistics exactly like F.batch_norm(training=True) does."""
    fusable = (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and bn.training and bn.affine
               and bn_supported(x.shape[1]) and act in ("softplus", "identity"))
    if not fusable:
        y = bn(x)
        return F.softplus(y) if act == "softplus" else (F.relu(y) if act == "relu" else y)
    track = bn.track_running_stats and bn.momentum is not None
    y, mean, rstd = _BnActFn.apply(x, bn.weight, bn.bias, bn.eps, 1 if act == "softplus" else 0,
                                   bn.running_mean if track else None, bn.running_var if track else None,
                                   bn.momentum if track else 0.0, bn.num_batches_tracked if track else None)
    if bn.track_running_stats and not track:          # cumulative moving average (momentum=None): rare
        with torch.no_grad():
            n = x.shape[0]
            bn.num_batches_tracked += 1
            mom = 1.0 / float(bn.num_batches_tracked)
            var_unbiased = (1.0 / (rstd * rstd) - bn.eps) * (n / max(n - 1, 1))
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var_unbiased, alpha=mom)
    return y


class _SsimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        lib = _native.ganet()
        shape = img1.shape
        H, W = shape[-2], shape[-1]
        planes = img1.numel() // (H * W)
        a = img1.contiguous().float()
        b = img2.contiguous().float()
        total = torch.empty(1, dtype=torch.float32, device=a.device)
        partials = torch.empty((3, planes, H, W), dtype=torch.float32, device=a.device)
        _native.ganet_check(lib.ganet_ssim_fwd(planes, H, W, _ptr(a), _ptr(b), _ptr(total), _ptr(partials),
                                               _stream(a.device)))
        ctx.save_for_backward(a, b, partials)
        ctx.shape = shape
        return total[0] / float(a.numel())

    @staticmethod
    def backward(ctx, g):
        lib = _native.ganet()
        a, b, partials = ctx.saved_tensors
        H, W = a.shape[-2], a.shape[-1]
        planes = a.numel() // (H * W)
        scale = (g / float(a.numel())).reshape(1).float().contiguous()
        d = torch.empty_like(a)
        _native.ganet_check(lib.ganet_ssim_bwd(planes, H, W, _ptr(a), _ptr(b), _ptr(partials), _ptr(scale),
                                               _ptr(d), _stream(a.device)))
        return d.reshape(ctx.shape), None


def ssim_mean(img1, img2):
    """Mean SSIM (window 11, sigma 1.5) of img1 vs img2 ([..., H, W]); differentiable w.r.t. img1."""
    return _SsimFn.apply(img1, img2)
