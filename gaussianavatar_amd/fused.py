"""Autograd wrappers of the fused feature-net / loss kernels (include/ganet.h).

Used by network.ShapeDecoder and losses.ssim whenever their tensors live on a HIP device; on
the CPU the same modules run the plain torch formulation (that is what the golden-vector tests
exercise), so the two formulations are checked against each other on the GPU box
(tests/test_fused_gpu.py).
"""
from __future__ import annotations

import ctypes
import weakref
import threading

import torch
import torch.nn.functional as F

from . import _dev, _native, parallel


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    """The current HIP stream of `device` as the void* every ganet entry point takes. Every launch passes through here, so
    this is also where the calling thread is (re-)bound to the module's profile object: the binding is per thread
    (include/ganet.h) and autograd runs backward on its own thread — one integer compare per call."""
    if getattr(_profile_tls, "mask", 0) != _profile_mask and _profile is not None:
        _native.ganet_check(_native.ganet().ganet_profile_bind(_profile, _profile_mask))
        _profile_tls.mask = _profile_mask
    return ctypes.c_void_p(_native.raw_stream(device))


_profiling = False        # per-kernel HIP events on: launches stay on ONE stream, so that a kernel's events time that kernel
_profile = None           # the GanetProfile object of this module (caller-owned, bound to the launching threads)
_profile_mask = 0
_profile_tls = threading.local()


def profile_kernels() -> tuple:
    lib = _native.ganet()
    return tuple(lib.ganet_profile_kernel_name(i).decode() for i in range(lib.ganet_profile_count()))


def profile_enable(on=True) -> None:
    """Bracket the decoder / SSIM kernel launches with HIP events on their stream (bench only). `on`:
    True = every kernel, False = off, or an iterable of kernel names to time only those."""
    names = profile_kernels()
    if on is True:
        mask = (1 << len(names)) - 1
    elif not on:
        mask = 0
    else:
        mask = sum(1 << names.index(k) for k in on)
    global _profiling, _profile, _profile_mask
    if _profile is None and mask:
        _profile = ctypes.c_void_p(_native.ganet().ganet_profile_create())
    _profile_mask = mask
    _profiling = bool(mask)


def profile_read(reset: bool = True) -> dict:
    """{kernel name: (total ms, launches)} from the in-library HIP events (blocks until they completed)."""
    lib = _native.ganet()
    n = lib.ganet_profile_count()
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int64 * n)()
    if _profile is not None:
        _native.ganet_check(lib.ganet_profile_read(_profile, ms, cnt, 1 if reset else 0))
    return {lib.ganet_profile_kernel_name(i).decode(): (ms[i], int(cnt[i])) for i in range(n)}


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


class _SsimL1Fn(torch.autograd.Function):
    """(mean SSIM, mean |img1 - img2|) in one pass over the images; one pass backward."""

    @staticmethod
    def forward(ctx, img1, img2):
        lib = _native.ganet()
        shape = img1.shape
        H, W = shape[-2], shape[-1]
        planes = img1.numel() // (H * W)
        a = img1.contiguous().float()
        b = img2.contiguous().float()
        sums = torch.empty(lib.ganet_ssim_sums_floats(), dtype=torch.float32, device=a.device)
        partials = torch.empty((3, planes, H, W), dtype=torch.float32, device=a.device)
        _native.ganet_check(lib.ganet_ssim_fwd(planes, H, W, _ptr(a), _ptr(b), 1.0 / float(a.numel()), _ptr(sums),
                                               _ptr(partials), _stream(a.device)))
        ctx.save_for_backward(a, b, partials)
        ctx.shape = shape
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g_ssim, g_l1):
        lib = _native.ganet()
        a, b, partials = ctx.saved_tensors
        H, W = a.shape[-2], a.shape[-1]
        planes = a.numel() // (H * W)
        scalar = lambda g: None if g is None else g.reshape(1).float().contiguous()
        gs, gl = scalar(g_ssim), scalar(g_l1)
        d = torch.empty_like(a)
        _native.ganet_check(lib.ganet_ssim_bwd(planes, H, W, _ptr(a), _ptr(b), _ptr(partials), 1.0 / float(a.numel()),
                                               _ptr(gs), _ptr(gl), _ptr(d), _stream(a.device)))
        return d.reshape(ctx.shape), None


def ssim_l1_mean(img1, img2):
    """(mean SSIM (window 11, sigma 1.5), mean absolute difference) of img1 vs img2 ([..., H, W]);
    differentiable w.r.t. img1."""
    return _SsimL1Fn.apply(img1, img2)


def ssim_mean(img1, img2):
    return _SsimL1Fn.apply(img1, img2)[0]


class _MeanSqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _native.ganet()
        # element order does not matter: any dense layout (channels-last parameters) is read in place
        xc = x if (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)) else x.contiguous()
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        _native.ganet_check(lib.ganet_mean_sq_fwd(xc.numel(), _ptr(xc), 1.0 / float(xc.numel()), _ptr(out),
                                                  _stream(x.device)))
        ctx.save_for_backward(xc)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = _native.ganet()
        (xc,) = ctx.saved_tensors
        dx = torch.empty_like(xc, memory_format=torch.preserve_format)
        _native.ganet_check(lib.ganet_mean_sq_bwd(xc.numel(), _ptr(xc), 1.0 / float(xc.numel()),
                                                  _ptr(g.reshape(1).float().contiguous()), _ptr(dx), _stream(xc.device)))
        return dx


def mean_sq(x):
    """mean(x ** 2) as one launch each way (float32 CUDA tensors)."""
    return _MeanSqFn.apply(x)


class _WeightedSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, bias, *terms):
        lib = _native.ganet()
        n = len(terms)
        ts = [t.reshape(1) if t.dtype == torch.float32 and t.is_contiguous() else t.reshape(1).float().contiguous()
              for t in terms]
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        w = (ctypes.c_float * n)(*[float(v) for v in weights])
        out = torch.empty(1, dtype=torch.float32, device=ts[0].device)
        _native.ganet_check(lib.ganet_weighted_sum_fwd(n, ptrs, w, float(bias), _ptr(out), _stream(out.device)))
        ctx.w, ctx.n = w, n
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = _native.ganet()
        d = torch.empty(ctx.n, dtype=torch.float32, device=g.device)
        _native.ganet_check(lib.ganet_weighted_sum_bwd(ctx.n, ctx.w, _ptr(g.reshape(1).float().contiguous()), _ptr(d),
                                                       _stream(g.device)))
        return (None, None) + tuple(d[i] for i in range(ctx.n))


def weighted_sum(terms, weights, bias: float = 0.0):
    """bias + sum_i weights[i] * terms[i] for zero-dimensional CUDA tensors and python-number weights:
    one launch forward, one backward."""
    return _WeightedSumFn.apply(tuple(weights), bias, *terms)


def bilinear_taps(Wm: torch.Tensor):
    """Dense [S,R] bilinear weight matrix (<= 2 non-zeros per row) -> the tap lists the up-sampling
    kernels take: (idx int32 [S,2], w float32 [S,2]) and the transposed lists in CSR form
    (ptr int32 [R+1], src int32 [nnz], w float32 [nnz])."""
    S, R = Wm.shape
    w, idx = Wm.topk(2, dim=1)
    assert float((Wm.sum(1) - w.sum(1)).abs().max()) < 1e-6, "more than two taps per row"
    idx32 = idx.to(torch.int32).contiguous()
    w = w.contiguous()
    ic, wc = idx.cpu().reshape(-1), w.cpu().reshape(-1)
    src = torch.arange(S).repeat_interleave(2)
    keep = wc != 0
    ic, wc, src = ic[keep], wc[keep], src[keep]
    order = torch.argsort(ic * S + src)
    ic, wc, src = ic[order], wc[order], src[order]
    ptr = torch.zeros(R + 1, dtype=torch.int64)
    ptr[1:] = torch.bincount(ic, minlength=R).cumsum(0)
    dev = Wm.device
    return (idx32, w, ptr.to(torch.int32).to(dev), src.to(torch.int32).to(dev), wc.float().to(dev))


class _UpsampleCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pix, uv, rows, cols, ldx):
        lib = _native.ganet()
        b, C, R, _ = pix.shape
        S = rows[0].shape[0]
        feat = pix.permute(0, 2, 3, 1).contiguous()                     # channels-last [b,R,R,C]
        uvc = uv.contiguous()
        x = torch.empty((b * S * S, ldx), dtype=torch.float32, device=pix.device)
        _native.ganet_check(lib.ganet_upsample_cat_fwd(b, S, R, C, _ptr(feat), _ptr(rows[0]), _ptr(rows[1]),
                                                       _ptr(cols[0]), _ptr(cols[1]), _ptr(uvc), _ptr(x), ldx,
                                                       _stream(pix.device)))
        ctx.rows, ctx.cols, ctx.dims = rows, cols, (b, C, R, S, ldx)
        return x

    @staticmethod
    def backward(ctx, dx):
        lib = _native.ganet()
        b, C, R, S, ldx = ctx.dims
        rows, cols = ctx.rows, ctx.cols
        dx = dx if (dx.stride(1) == 1 and dx.stride(0) >= C) else dx.contiguous()
        dfeat = torch.empty((b, R, R, C), dtype=torch.float32, device=dx.device)
        tmp = torch.empty((b, S, R, C), dtype=torch.float32, device=dx.device)
        _native.ganet_check(lib.ganet_upsample_cat_bwd(b, S, R, C, _ptr(dx), dx.stride(0), _ptr(rows[2]), _ptr(rows[3]),
                                                       _ptr(rows[4]), _ptr(cols[2]), _ptr(cols[3]), _ptr(cols[4]),
                                                       _ptr(tmp), _ptr(dfeat), _stream(dx.device)))
        return dfeat.permute(0, 3, 1, 2), None, None, None, None


def upsample_cat(pix, uv, rows, cols, ldx):
    """pix [b,64,R,R] bilinearly up-sampled at the separable texel grid (rows/cols = bilinear_taps of the
    two weight matrices), concatenated with uv [b,S*S,2] and zero-padded to ldx columns -> [b*S*S, ldx]."""
    return _UpsampleCatFn.apply(pix, uv, rows, cols, ldx)


def geom_convs_supported(x, weights) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 64 and x.shape[3] % 64 == 0
            and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 24) and 0 < len(weights) <= 4
            and all(tuple(w.shape) == (64, 64, 5, 5) and w.dtype == torch.float32 for w in weights))


class _GeomConvFn(torch.autograd.Function):
    """A chain of 5x5 / pad 2 / bias-free 64 -> 64 convolutions (GeomConvLayers) on the hand-written kernels of
    csrc/ganet_conv.hip. Input: logical NCHW; the result is a logical-NCHW VIEW of a channels-last buffer — what
    the up-sampling kernel reads — so no layout copy follows."""

    @staticmethod
    def forward(ctx, x, *weights):
        lib = _native.ganet()
        b, _, H, W = x.shape
        n = len(weights)
        st = _stream(x.device)
        xh = x.permute(0, 2, 3, 1).contiguous()
        ws = [w.contiguous() for w in weights]
        packed = torch.empty(lib.ganet_conv5_packed_bytes(n), dtype=torch.uint8, device=x.device)
        ptrs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
        _native.ganet_check(lib.ganet_conv5_pack(n, ptrs, _ptr(packed), st))
        maps = [xh]
        for i in range(n):
            y = torch.empty_like(xh)
            _native.ganet_check(lib.ganet_conv5_apply(b, H, W, _ptr(maps[-1]), _ptr(packed), i, 0, _ptr(y), st))
            maps.append(y)
        ctx.save_for_backward(packed, *maps[:-1])
        ctx.dims = (b, H, W, n)
        return maps[-1].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _native.ganet()
        packed, *maps = ctx.saved_tensors
        b, H, W, n = ctx.dims
        st = _stream(g.device)
        gh = g.permute(0, 2, 3, 1).contiguous()
        nbytes = lib.ganet_conv5_wgrad_workspace(b, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        dws = [None] * n
        for i in reversed(range(n)):
            if ctx.needs_input_grad[1 + i]:
                dws[i] = torch.empty((64, 64, 5, 5), dtype=torch.float32, device=g.device)
                _native.ganet_check(lib.ganet_conv5_wgrad(b, H, W, _ptr(maps[i]), _ptr(gh), _ptr(dws[i]), _ptr(ws),
                                                          nbytes, st))
            if i > 0 or ctx.needs_input_grad[0]:
                gx = torch.empty_like(gh)
                _native.ganet_check(lib.ganet_conv5_apply(b, H, W, _ptr(gh), _ptr(packed), i, 1, _ptr(gx), st))
                gh = gx
        dx = gh.permute(0, 3, 1, 2) if ctx.needs_input_grad[0] else None
        return (dx,) + tuple(dws)


def geom_convs(x, weights):
    """conv5x5(... conv5x5(x, weights[0]) ..., weights[-1]) (padding 2, no bias) for x [b,64,H,W]."""
    return _GeomConvFn.apply(x, *weights)


class _DecodePackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, res, s_logit, c_logit, valid_index, inv_index, res_scale, scale_mult):
        lib = _native.ganet()
        b, HW = res.shape[0], res.shape[1]
        N = valid_index.shape[0]
        res, s_logit, c_logit = res.contiguous(), s_logit.contiguous(), c_logit.contiguous()
        flat = torch.empty(b * N * 7, dtype=torch.float32, device=res.device)
        sums = torch.empty(2, dtype=torch.float32, device=res.device)
        norms = (1.0 / float(b * HW * 3), 1.0 / float(max(b * N, 1)))
        _native.ganet_check(lib.ganet_decode_pack_fwd(b, HW, N, _ptr(res), _ptr(s_logit), _ptr(c_logit),
                                                      _ptr(valid_index), float(res_scale), float(scale_mult),
                                                      norms[0], norms[1], _ptr(flat), _ptr(sums), _stream(res.device)))
        ctx.save_for_backward(res, s_logit, c_logit, inv_index)
        ctx.consts = (float(res_scale), float(scale_mult), N, norms)
        return flat, sums[0], sums[1]

    @staticmethod
    def backward(ctx, d_flat, d_sq, d_scale):
        lib = _native.ganet()
        res, s_logit, c_logit, inv_index = ctx.saved_tensors
        res_scale, scale_mult, N, norms = ctx.consts
        b, HW = res.shape[0], res.shape[1]
        d_res, d_s, d_c = torch.empty_like(res), torch.empty_like(s_logit), torch.empty_like(c_logit)
        scalar = lambda g: None if g is None else g.reshape(1).float().contiguous()
        d_sq, d_scale = scalar(d_sq), scalar(d_scale)
        d_flat = None if d_flat is None else d_flat.contiguous()
        _native.ganet_check(lib.ganet_decode_pack_bwd(b, HW, N, _ptr(res), _ptr(s_logit), _ptr(c_logit), _ptr(inv_index),
                                                      res_scale, scale_mult, norms[0], norms[1], _ptr(d_flat),
                                                      _ptr(d_sq), _ptr(d_scale), _ptr(d_res), _ptr(d_s), _ptr(d_c),
                                                      _stream(res.device)))
        return d_res, d_s, d_c, None, None, None, None


def decode_pack(res, s_logit, c_logit, valid_index, inv_index, res_scale, scale_mult):
    """Decoder-head logits ([b,HW,3], [b,HW,1], [b,HW,3]) -> (flat [b*N*7] = residual*res_scale [b,N,3] |
    sigmoid(scale)*scale_mult [b,N] | sigmoid(colour) [b,N,3] on the valid texels (split_records gives
    the views); mean over all texels of (res_scale*residual)^2; mean of the valid texels' scales)."""
    return _DecodePackFn.apply(res, s_logit, c_logit, valid_index, inv_index, res_scale, scale_mult)


class _SplitRecords(torch.autograd.Function):
    """flat [b*N*7] -> contiguous views (residual [b,N,3], scale [b,N,1], colour [b,N,3]). The backward
    assembles the flat gradient with one cat instead of three zero-filled slice gradients and their sum."""

    @staticmethod
    def forward(ctx, flat, b, N):
        ctx.dims = (b, N)
        return (flat[:b * N * 3].view(b, N, 3), flat[b * N * 3:b * N * 4].view(b, N, 1),
                flat[b * N * 4:].view(b, N, 3))

    @staticmethod
    def backward(ctx, g_res, g_scale, g_col):
        b, N = ctx.dims
        ref = next(g for g in (g_res, g_scale, g_col) if g is not None)
        z = lambda c: ref.new_zeros(b * N * c)
        parts = [g.reshape(-1) if g is not None else z(c) for g, c in ((g_res, 3), (g_scale, 1), (g_col, 3))]
        return torch.cat(parts), None, None


def split_records(flat, b, N):
    return _SplitRecords.apply(flat, b, N)


class _ExpandRecords(torch.autograd.Function):
    """flat [b*N*7] -> (residual [B,N,3], scale [B,N,3], colour [B,N,3]): the record views of split_records broadcast
    over the B frames (b = 1) and the scalar scale over three axes — what avatar_model hands to skinning and to the
    rasterizer. The backward is ONE kernel (sum over frames / over the scale's copies, written straight into the flat
    gradient) instead of autograd's four expand-backward reductions, a cat and its fills."""

    @staticmethod
    def forward(ctx, flat, b, N, B):
        ctx.dims = (b, N, B)
        res = flat[:b * N * 3].view(b, N, 3).expand(B, -1, -1)
        scale = flat[b * N * 3:b * N * 4].view(b, N, 1).expand(B, -1, 3)
        col = flat[b * N * 4:].view(b, N, 3).expand(B, -1, -1)
        return res, scale, col

    @staticmethod
    def backward(ctx, g_res, g_scale, g_col):
        b, N, B = ctx.dims
        ref = next(g for g in (g_res, g_scale, g_col) if g is not None)
        c = lambda g: None if g is None else g.contiguous().float()
        g_res, g_scale, g_col = c(g_res), c(g_scale), c(g_col)
        d = torch.empty(b * N * 7, dtype=torch.float32, device=ref.device)
        _native.ganet_check(_native.ganet().ganet_records_bwd(b, B, N, _ptr(g_res), _ptr(g_scale), _ptr(g_col), _ptr(d),
                                                            _stream(ref.device)))
        return d, None, None, None


def expand_records(flat, b, N, B):
    """(residual, scale x3, colour) [B,N,3] views of the packed records; see _ExpandRecords. b is 1 (records shared by
    the frames) or B."""
    if flat.is_cuda:
        return _ExpandRecords.apply(flat, b, N, B)
    res, s1, col = split_records(flat, b, N)
    if b != B:
        res, s1, col = (t.expand(B, -1, -1) for t in (res, s1, col))
    return res, s1.expand(-1, -1, 3), col


# ------------------------------------------------------------------------------------------------
# Whole-decoder autograd function on the fused layer kernels (ganet_mlp.hip): no normalised
# activation is ever stored, BatchNorm statistics come out of the producing GEMM's epilogue.

_TRUNK = (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"), ("conv4", "bn4"), ("conv5", "bn5"))
_HEAD_TAGS = ("", "N", "SH")
_K1_PAD = 72          # decoder input (64 geometry features + 2 uv = 66 columns) padded to 8-float blocks


def decoder_input_pad(dec, like) -> int:
    """Zero columns a caller may append to the decoder input so that the fused path can use it without
    a padding copy (0 when the fused path does not apply)."""
    probe = like.new_empty((1, dec.in_size))
    return _K1_PAD - dec.in_size if decoder_supported(dec, probe) else 0


# Per-decoder caches live beside the modules, not inside them: a module must stay deep-copyable and picklable (a ctypes
# struct of pointers is neither), and the entries die with the module.
_module_cache = weakref.WeakKeyDictionary()
_params_cache = weakref.WeakKeyDictionary()


def _decoder_modules(dec):
    """(conv, bn) module pairs of the decoder's 11 BatchNorm layers and its three output convolutions, looked up ONCE per
    decoder object: nn.Module.__getattr__ walks _parameters / _buffers / _modules on every access, and the hot path asked
    for these ~130 times per iteration (65 us in decoder_supported, 45 us in the parameter list: tools/prof_enqueue.py)."""
    cached = _module_cache.get(dec)
    sig = tuple(map(id, dec._modules.values()))          # a replaced submodule (dec.bn3 = ...) invalidates the entry
    if cached is None or cached[2] != sig:
        pairs = [(getattr(dec, c), getattr(dec, b)) for c, b in _decoder_bn_layers()]
        outs = [getattr(dec, f"conv8{t}") for t in _HEAD_TAGS]
        cached = _module_cache[dec] = (pairs, outs, sig)
    return cached


def decoder_supported(dec, x) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and not dec.use_relu
            and dec.hsize == 128 and dec.in_size <= _K1_PAD and x.shape[1] in (dec.in_size, _K1_PAD)):
        return False
    training = dec.training
    for _, bn in _decoder_modules(dec)[0]:
        if bn.training != training or not bn.affine or bn.momentum is None:
            return False
    return training or not torch.is_grad_enabled()


def _decoder_bn_layers():
    out = list(_TRUNK)
    for t in _HEAD_TAGS:
        out += [(f"conv6{t}", f"bn6{t}"), (f"conv7{t}", f"bn7{t}")]
    return out


# hidden 128 -> 128 layers: data gradient + weight gradient in ONE pass over the activations
# (ganet_mlp_bwd_fused, csrc/ganet_layer_bwd.hip): 4 instead of 7 [M,128] tensors through HBM. Used whenever the
# row count is a multiple of 32; the separate kernels remain for ragged row counts and for the layers whose input
# is the 72-column decoder input. (Tests switch it off to compare the two formulations.)
_FUSED_BWD = _dev.knobs.one_pass_backward


# The weight-gradient launches of the decoder backward are off its dependency chain (nothing reads dW before the
# batched reduction at the end): they are issued on a side stream (GA_DEV=wgrad_stream=0: on the main one), ordered by
# events behind the statistics they need, so that their heads (weight staging, launch latency) fill the tails of the
# data-gradient kernels. LDS keeps the two kernel families from sharing a CU, so they do not slow each other's main
# loops. Measured: +1.3 % iterations/s.
_WGRAD_STREAM = _dev.knobs.wgrad_stream
_side_streams = {}
_encoder_streams = {}


def encoder_stream(device):
    """the pose encoder's own stream (one per device; distinct from the weight-gradient side stream, which the encoder's
    backward uses itself)"""
    key = torch.device(device).index or 0
    st = _encoder_streams.get(key)
    if st is None:
        st = _encoder_streams[key] = (torch.cuda.Stream(device=device), torch.cuda.Event())
    return st[0]


def encoder_event(device):
    encoder_stream(device)
    return _encoder_streams[torch.device(device).index or 0][1]




def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class _RowSweep:
    """Alternates GANET_ROWS_UP / GANET_ROWS_DOWN between consecutive big launches of a pass (include/ganet.h:
    each kernel starts on the rows its predecessor touched last, which are still in the Infinity Cache)."""
    enabled = _dev.knobs.row_sweep

    def __init__(self):
        self.k = 0

    def next(self) -> int:
        if not self.enabled:
            return 0
        self.k += 1
        return 1 if self.k % 2 else 2


def _mlp_fwd(lib, M, N, x1, x2, scale, shift, W, bias, col_part, dev, row_order=0, stat_shift=None):
    z = torch.empty((M, N), dtype=torch.float32, device=dev)
    K1 = 0 if x1 is None else x1.shape[1]
    K2 = 0 if x2 is None else x2.shape[1]
    _native.ganet_check(lib.ganet_mlp_fwd(
        M, N, K1, K2, _ptr(x1), 0 if x1 is None else x1.stride(0), _ptr(x2), 0 if x2 is None else x2.stride(0),
        _ptr(scale), _ptr(shift), _ptr(W), _ptr(bias), _ptr(z), z.stride(0), _ptr(col_part), _ptr(stat_shift),
        row_order, _stream(dev)))
    return z


# The whole decoder as one native call each way (csrc/ganet_decoder.hip: the same launch sequence as _DecoderFn below,
# issued from C). Training mode, single-rank BatchNorm statistics, row count a multiple of 32; everything else — and the
# tests that compare the two — takes the per-layer path. (Tests switch it off to exercise the per-layer path.)
_NATIVE_DECODER = _dev.knobs.native_decoder


def _native_decoder_ok(dec, M, sync) -> bool:
    if not (_NATIVE_DECODER and _FUSED_BWD and dec.training and not sync and M % 32 == 0):
        return False
    tracks = [bn.track_running_stats for _, bn in _decoder_modules(dec)[0]]
    return all(tracks) or not any(tracks)


def _native_decoder_params(dec, params, nl):
    """GanetDecoderParams for the module's current tensors (and the tensors it points at, to keep them alive). The struct
    is rebuilt only when a tensor moved: the optimiser updates parameters in place, so an iteration re-uses the previous
    one's (75 us per build, twice per iteration)."""
    bns = [bn for _, bn in _decoder_modules(dec)[0]]
    # everything the struct points at or copies: parameters, the three BatchNorm buffers, eps / momentum (a
    # reassigned running_var / num_batches_tracked or a changed momentum must rebuild it)
    key = tuple([p.data_ptr() for p in params] +
                [v for bn in bns for v in ((bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                            bn.num_batches_tracked.data_ptr()) if bn.track_running_stats else (0, 0, 0))] +
                [v for bn in bns for v in (float(bn.eps), float(bn.momentum))])
    cached = _params_cache.get(dec)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    P, keep = _build_decoder_params(dec, params, nl)
    if all(p.is_contiguous() for p in params):       # (a contiguous COPY of a strided weight would go stale in the cache)
        _params_cache[dec] = (key, P, keep)
    return P, keep


def _build_decoder_params(dec, params, nl):
    from ._native import GanetDecoderParams
    P = GanetDecoderParams()
    P.cin = dec.in_size
    layers = _decoder_bn_layers()
    keep = []
    for i, (_, bn_name) in enumerate(layers):
        bn = getattr(dec, bn_name)
        w, b, ga, be = (params[4 * i + k] for k in range(4))
        w = w if w.is_contiguous() else w.contiguous()
        keep += [w, b, ga, be]
        P.W[i], P.bias[i], P.gamma[i], P.beta[i] = w.data_ptr(), b.data_ptr(), ga.data_ptr(), be.data_ptr()
        if bn.track_running_stats:
            P.running_mean[i], P.running_var[i] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            P.num_batches_tracked[i] = bn.num_batches_tracked.data_ptr()
            keep += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
        P.eps[i], P.momentum[i] = float(bn.eps), float(bn.momentum)
    for j in range(3):
        w, b = params[4 * nl + 2 * j], params[4 * nl + 2 * j + 1]
        w = w if w.is_contiguous() else w.contiguous()
        keep += [w, b]
        P.W8[j], P.b8[j], P.n8[j] = w.data_ptr(), b.data_ptr(), w.shape[0]
    return P, keep


class _DecoderFn(torch.autograd.Function):
    """(x [M,in], flat parameter list) -> (residual [M,3], scale logits [M,1], colour logits [M,3]).
    Parameter order: for every (conv, bn) of _decoder_bn_layers(): conv.weight, conv.bias,
    bn.weight, bn.bias; then conv8{tag}.weight, conv8{tag}.bias per head."""

    @staticmethod
    def forward(ctx, x, dec, m_global, *params):
        """m_global: None, or the row count of the GLOBAL batch when this rank holds only its share of the
        rows (texel-sharded stage 1, frame-sharded stage 2): the BatchNorm column sums of every layer are
        then all-reduced, so statistics (and their backward) are those of the single-process batch."""
        lib = _native.ganet()
        dev = x.device
        M, cin = x.shape[0], dec.in_size
        sync = m_global is not None and int(m_global) != M
        Mg = int(m_global) if sync else M
        ctx.sync, ctx.Mg = sync, Mg
        layers = _decoder_bn_layers()
        nl = len(layers)
        conv_w = [params[4 * i].squeeze(-1) for i in range(nl)]
        conv_b = [params[4 * i + 1] for i in range(nl)]
        gammas = [params[4 * i + 2] for i in range(nl)]
        betas = [params[4 * i + 3] for i in range(nl)]
        out_w = [params[4 * nl + 2 * j].squeeze(-1) for j in range(3)]
        out_b = [params[4 * nl + 2 * j + 1] for j in range(3)]
        training = dec.training
        if x.shape[1] == _K1_PAD and x.is_contiguous():
            xp = x                           # the caller appended the zero columns (decoder_input_pad)
        else:
            xp = torch.zeros((M, _K1_PAD), dtype=torch.float32, device=dev)
            xp[:, :cin] = x
        ctx.x_cols = x.shape[1]
        ctx.native = _native_decoder_ok(dec, M, sync)
        if ctx.native:
            P, keep = _native_decoder_params(dec, params, nl)
            saved = torch.empty(lib.ganet_decoder_saved_floats(M), dtype=torch.float32, device=dev)
            outs = [torch.empty((M, P.n8[j]), dtype=torch.float32, device=dev) for j in range(3)]
            wsb = lib.ganet_decoder_fwd_workspace()
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            optrs = (ctypes.c_void_p * 3)(*[o.data_ptr() for o in outs])
            _native.ganet_check(lib.ganet_decoder_fwd(M, _ptr(xp), ctypes.byref(P), _ptr(saved), optrs, _ptr(ws), wsb,
                                                      _stream(dev)))
            ctx.dec, ctx.cin, ctx.nl = dec, cin, nl
            ctx.save_for_backward(xp, saved, *params)
            return tuple(outs)
        pad_w = lambda w: torch.cat([w, w.new_zeros(w.shape[0], _K1_PAD - cin)], 1).contiguous()
        col_part = torch.empty(lib.ganet_mlp_stats_floats(128), dtype=torch.float32, device=dev) if training else None

        zs, stats = [], []            # per BN layer: pre-activation, (mean, rstd, scale, shift)
        sweep = _RowSweep()

        # Synchronised statistics all-reduce sums of (z - shift): every rank must subtract the SAME shift. The running
        # means are equal on all ranks as long as the replicas were synchronised and stepped together — rather than
        # rely on that, rank 0's are broadcast (one 11 x 128 message per forward pass) and used as the shifts.
        sync_shift = None
        if sync and training and all(getattr(dec, bn).track_running_stats for _, bn in layers):
            sync_shift = torch.stack([getattr(dec, bn).running_mean for _, bn in layers]).contiguous()
            parallel.broadcast_(sync_shift)

        def stat_shift_of(i):
            """BatchNorm statistics are accumulated about the layer's running mean (sum (z - s), sum (z - s)^2):
            raw fp32 sums would cancel in E[z^2] - mean^2 once |mean| >> std."""
            if sync_shift is not None:
                return sync_shift[i]
            bn = getattr(dec, layers[i][1])
            return bn.running_mean if (training and bn.track_running_stats) else None

        def bn_stats(i, cp):
            """statistics of layer i from its column-sum partials `cp` (already summed over the ranks when sync)"""
            bn = getattr(dec, layers[i][1])
            if training:
                mean, rstd, sc, sh = (torch.empty(128, dtype=torch.float32, device=dev) for _ in range(4))
                track = bn.track_running_stats
                _native.ganet_check(lib.ganet_mlp_stats(
                    Mg, 128, _ptr(cp), _ptr(gammas[i]), _ptr(betas[i]), float(bn.eps), _ptr(mean), _ptr(rstd),
                    _ptr(sc), _ptr(sh), _ptr(bn.running_mean if track else None),
                    _ptr(bn.running_var if track else None), float(bn.momentum),
                    _ptr(bn.num_batches_tracked if track else None), _ptr(stat_shift_of(i)), _stream(dev)))
            else:
                mean = bn.running_mean
                rstd = torch.rsqrt(bn.running_var + bn.eps)
                sc = (gammas[i] * rstd).contiguous()
                sh = (betas[i] - mean * sc).contiguous()
            return mean, rstd, sc, sh

        # column-sum partials: one buffer per layer of a LEVEL (the three heads' conv6 / conv7 are independent of
        # each other): with synchronised statistics a level's layers share ONE all-reduce (11 -> 7 messages per pass)
        col_parts = [col_part] + ([torch.empty_like(col_part) for _ in range(2)] if (training and sync) else [])

        def hidden_level(items):
            """items: [(layer i, x1, W, src)], layers that do not depend on each other: z = [x1 | act(src)] W^T + b for
            each, then their BatchNorm statistics (column sums all-reduced in one message when sync)."""
            cps = []
            for k, (i, x1, W, src) in enumerate(items):
                x2 = sc = sh = None
                if src is not None:
                    x2, (_, _, sc, sh) = zs[src], stats[src]
                cp = col_parts[k] if (training and sync) else col_part
                while len(zs) <= i:
                    zs.append(None)
                    stats.append(None)
                zs[i] = _mlp_fwd(lib, M, 128, x1, x2, sc, sh, W, conv_b[i], cp, dev, sweep.next(), stat_shift_of(i))
                cps.append(cp)
                if not (training and sync):
                    stats[i] = bn_stats(i, cp)
            if training and sync:
                _allreduce_partials(cps, 256)
                for (i, _x1, _W, _src), cp in zip(items, cps):
                    stats[i] = bn_stats(i, cp)

        def hidden(i, x1, W, src):
            hidden_level([(i, x1, W, src)])

        w1p = pad_w(conv_w[0])
        hidden(0, xp, w1p, None)
        for i in (1, 2, 3):
            hidden(i, None, conv_w[i].contiguous(), i - 1)
        w5 = conv_w[4]
        w5p = torch.cat([pad_w(w5[:, :cin]), w5[:, cin:]], 1).contiguous()      # [128, 72 + 128]
        hidden(4, xp, w5p, 3)
        outs = []
        if training and sync:      # the three heads level by level: one statistics message per level
            hidden_level([(5 + 2 * j, None, conv_w[5 + 2 * j].contiguous(), 4) for j in range(3)])
            hidden_level([(6 + 2 * j, None, conv_w[6 + 2 * j].contiguous(), 5 + 2 * j) for j in range(3)])
        for j in range(3):
            i6, i7 = 5 + 2 * j, 6 + 2 * j
            if not (training and sync):
                hidden(i6, None, conv_w[i6].contiguous(), 4)
                hidden(i7, None, conv_w[i7].contiguous(), i6)
            _, _, sc, sh = stats[i7]
            outs.append(_mlp_fwd(lib, M, out_w[j].shape[0], None, zs[i7], sc, sh, out_w[j].contiguous(),
                                 out_b[j], None, dev, sweep.next()))
        ctx.dec = dec
        ctx.cin = cin
        ctx.nl = nl
        flat_stats = [t for st in stats for t in st]
        ctx.save_for_backward(xp, *zs, *flat_stats, *conv_w, *gammas, *betas, *out_w)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d_outs):
        """No dL/dy or dz tensor is materialised: every hidden layer i keeps G_i = dL/dy_i . softplus'(u_i)
        (written by the producing kernel's epilogue) and the per-column coefficients (A, q, p) with
        dz_i = A G_i + q z_i + p; the data-gradient and weight-gradient kernels assemble dz_i on load
        (include/ganet.h, ganet_mlp_bwd.hip)."""
        lib = _native.ganet()
        nl, cin = ctx.nl, ctx.cin
        sv = ctx.saved_tensors
        if ctx.native:
            # the one-call path back-propagates all three heads; a head the objective does not use (autograd.grad on
            # a partial objective, set_materialize_grads(False)) contributes a zero gradient
            return _DecoderFn._backward_native(ctx, lib, sv, d_outs)
        xp = sv[0]
        zs = sv[1:1 + nl]
        fs = sv[1 + nl:1 + 5 * nl]
        stats = [fs[4 * i:4 * i + 4] for i in range(nl)]          # mean, rstd, scale, shift
        conv_w = sv[1 + 5 * nl:1 + 6 * nl]
        out_w = sv[1 + 8 * nl:1 + 8 * nl + 3]
        dev = xp.device
        M = xp.shape[0]
        st = _stream(dev)
        g_conv_w, g_conv_b, g_gamma, g_beta = [None] * nl, [None] * nl, [None] * nl, [None] * nl
        g_out_w, g_out_b = [None] * 3, [None] * 3
        # every weight gradient leaves its per-workgroup partial sums in its own slice of one
        # workspace; ONE launch reduces all of them at the end (they are off the dependency chain)
        wg_bytes = (lib.ganet_wgrad_act_workspace(M, 128, 128) + 255) // 256 * 256
        max_jobs = 16
        wg_ws = torch.empty(wg_bytes * max_jobs, dtype=torch.uint8, device=dev)
        jobs = (_native.GanetWgradJob * max_jobs)()
        njobs = [0]
        sweep = _RowSweep()
        n_data, n_head = lib.ganet_mlp_bwd_data_parts(), lib.ganet_mlp_head_bwd_parts()
        col_part = torch.empty(max(n_data, n_head) * 256, dtype=torch.float32, device=dev)
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)

        main_stream = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if (_WGRAD_STREAM and not _profiling) else None
        wst = st if side is None else ctypes.c_void_p(side.cuda_stream)
        if side is not None:
            wg_ws.record_stream(side)

        def wgrad(g, gi, src, K=128):
            """dW [N,K], db [N]. g operand: raw tensor (gi None) or layer gi's (G, z, coef) triple;
            x operand: act(bn(zs[src])), or the padded decoder input when src is None."""
            if gi is None:
                gt, gz, coef, N = g, None, None, g.shape[1]
            else:
                gt, gz, coef, N = Gs[gi], zs[gi], coefs[gi], 128
            x = xp if src is None else zs[src]
            sc, sh = (None, None) if src is None else stats[src][2:]
            dW, db = f32(N, K), f32(N)
            j = njobs[0]
            ws = wg_ws.data_ptr() + j * wg_bytes
            if side is not None:
                # everything this launch reads has been enqueued on the main stream by now
                side.wait_stream(main_stream)
                for t in (gt, gz, coef, x, sc, sh, dW, db):
                    if t is not None:
                        t.record_stream(side)
            _native.ganet_check(lib.ganet_wgrad_act(
                M, N, K, _ptr(gt), gt.stride(0), _ptr(gz), 0 if gz is None else gz.stride(0), _ptr(coef),
                _ptr(x), x.stride(0), _ptr(sc), _ptr(sh), None, None, ws, wg_bytes, sweep.next(), wst))
            jobs[j].workspace, jobs[j].M, jobs[j].N, jobs[j].K = ws, M, N, K
            jobs[j].dW, jobs[j].db, jobs[j].nblocks = dW.data_ptr(), db.data_ptr(), 0
            njobs[0] = j + 1
            return dW, db          # filled by the batched reduction at the end of backward

        def finish(i, nparts, cp=None, reduced=False):
            """column sums of (G_i, G_i z_i) (partials `cp`, default col_part) -> coefficients of layer i, d gamma_i,
            d beta_i. reduced: the partials were already summed over the ranks (level-batched all-reduce)."""
            cp = col_part if cp is None else cp
            mean, rstd, sc, _ = stats[i]
            coef, dg, dbt = f32(3 * 128), f32(128), f32(128)
            if ctx.sync and not reduced:
                _allreduce_partials([cp[:nparts * 256]], 256)
            _native.ganet_check(lib.ganet_mlp_bwd_stats(ctx.Mg, nparts, _ptr(cp), _ptr(mean), _ptr(rstd), _ptr(sc),
                                                        _ptr(coef), _ptr(dg), _ptr(dbt), st))
            if ctx.sync:
                # d gamma / d beta come out of the GLOBAL sums, i.e. complete on every rank, while the other
                # parameter gradients are partial sums over the rank's rows: pre-divide, the cross-rank
                # reduction of the step (sum, or average of per-rank objectives) then restores them
                dg, dbt = dg / parallel.world_size(), dbt / parallel.world_size()
            coefs[i], g_gamma[i], g_beta[i] = coef, dg, dbt

        def data_grad(gi, W, out, accumulate, src, cp=None):
            """out[:, :O] (+)= dz_gi . W (W [128, O], possibly a column slice of a wider weight); with
            src: out = G_src (and its column sums in cp, default col_part)."""
            cp = col_part if cp is None else cp
            O = W.shape[1]
            sz = None if src is None else zs[src]
            sc, sh = (None, None) if src is None else stats[src][2:]
            _native.ganet_check(lib.ganet_mlp_bwd_data(
                M, O, _ptr(Gs[gi]), Gs[gi].stride(0), _ptr(zs[gi]), zs[gi].stride(0), _ptr(coefs[gi]), _ptr(W),
                W.stride(0), _ptr(out), out.stride(0), int(accumulate), _ptr(sz), 0 if sz is None else sz.stride(0),
                _ptr(sc), _ptr(sh), _ptr(cp) if src is not None else None, sweep.next(), st))

        fuse_ok = _FUSED_BWD and M % 32 == 0 and wg_bytes >= lib.ganet_mlp_bwd_fused_workspace()
        n_fused = lib.ganet_mlp_bwd_fused_parts()
        if fuse_ok and n_fused * 256 > col_part.numel():
            col_part = torch.empty(n_fused * 256, dtype=torch.float32, device=dev)

        def layer_bwd(i, src, W=None, out=None, accumulate=False, act=True, cp=None):
            """hidden layer i (input = act(bn(zs[src])) times W [128 out, 128 in], default conv_w[i]): d weight,
            d bias, and out (+)= dz_i . W — with act: out = G of the source layer, its column sums in cp (col_part).
            One pass over the activations when the fused kernel applies, else weight gradient + data gradient.
            Returns (dW, db, number of col_part rows)."""
            W = conv_w[i] if W is None else W
            cp = col_part if cp is None else cp
            if out is None:
                out = Gs[src] = f32(M, 128)
            if not fuse_ok:
                dW, db = wgrad(None, i, src)
                data_grad(i, W, out, accumulate, src if act else None, cp)
                return dW, db, n_data
            dW, db = f32(128, 128), f32(128)
            j = njobs[0]
            ws = wg_ws.data_ptr() + j * wg_bytes
            assert W.stride(1) == 1
            _native.ganet_check(lib.ganet_mlp_bwd_fused(
                M, _ptr(Gs[i]), _ptr(zs[i]), _ptr(coefs[i]), _ptr(W), W.stride(0), _ptr(out), int(accumulate),
                _ptr(zs[src]), _ptr(stats[src][2]), _ptr(stats[src][3]), int(act), _ptr(cp), ws, wg_bytes,
                sweep.next(), st))
            jobs[j].workspace, jobs[j].M, jobs[j].N, jobs[j].K = ws, M, 128, 128
            jobs[j].dW, jobs[j].db, jobs[j].nblocks = dW.data_ptr(), db.data_ptr(), n_fused
            njobs[0] = j + 1
            return dW, db, n_fused

        Gs, coefs = [None] * nl, [None] * nl
        heads = [j for j in range(3) if d_outs[j] is not None]
        G5 = f32(M, 128) if heads else None
        # With synchronised statistics the heads run level by level, so that a level's column sums travel in ONE
        # all-reduce (22 -> 14 messages per iteration together with the forward pass); without, head after head.
        level_order = ctx.sync and len(heads) > 1
        cps = [col_part] + ([torch.empty_like(col_part) for _ in range(len(heads) - 1)] if level_order else [])
        cp_of = (lambda pos: cps[pos]) if level_order else (lambda pos: col_part)

        def head_level(pos, j):          # conv8 of head j: G_7 and its column sums
            i7 = 6 + 2 * j
            g = d_outs[j].contiguous()
            N8 = g.shape[1]
            Gs[i7] = f32(M, 128)
            _, _, sc7, sh7 = stats[i7]
            dW, db = wgrad(g, None, i7)
            g_out_w[j], g_out_b[j] = dW.unsqueeze(-1), db
            _native.ganet_check(lib.ganet_mlp_head_bwd(M, N8, _ptr(g), _ptr(out_w[j].contiguous()), _ptr(zs[i7]),
                                                       zs[i7].stride(0), _ptr(sc7), _ptr(sh7), _ptr(Gs[i7]),
                                                       Gs[i7].stride(0), _ptr(cp_of(pos)), None, st))

        def conv7_level(pos, j):         # conv7 of head j: G_6 and its column sums; returns the partial-row count
            i6, i7 = 5 + 2 * j, 6 + 2 * j
            dW, db, nparts = layer_bwd(i7, i6, cp=cp_of(pos))
            g_conv_w[i7], g_conv_b[i7] = dW.unsqueeze(-1), db
            Gs[i7] = None
            return nparts

        def conv6_level(pos, j):         # conv6 of head j into G_5 (accumulating; the last one applies softplus')
            i6 = 5 + 2 * j
            last = pos == len(heads) - 1
            dW, db, nparts5 = layer_bwd(i6, 4, out=G5, accumulate=pos > 0, act=last)
            g_conv_w[i6], g_conv_b[i6] = dW.unsqueeze(-1), db
            Gs[i6] = None
            return nparts5

        if level_order:
            for pos, j in enumerate(heads):
                head_level(pos, j)
            _allreduce_partials([cps[pos][:n_head * 256] for pos in range(len(heads))], 256)
            for pos, j in enumerate(heads):
                finish(6 + 2 * j, n_head, cps[pos], reduced=True)
            nps = [conv7_level(pos, j) for pos, j in enumerate(heads)]
            _allreduce_partials([cps[pos][:nps[pos] * 256] for pos in range(len(heads))], 256)
            for pos, j in enumerate(heads):
                finish(5 + 2 * j, nps[pos], cps[pos], reduced=True)
            for pos, j in enumerate(heads):
                nparts5 = conv6_level(pos, j)
        else:
            for pos, j in enumerate(heads):
                head_level(pos, j)
                finish(6 + 2 * j, n_head)
                nparts = conv7_level(pos, j)
                finish(5 + 2 * j, nparts)
                nparts5 = conv6_level(pos, j)
        dx = None
        if heads:
            Gs[4] = G5
            finish(4, nparts5)
            w5 = conv_w[4]
            dWx, _ = wgrad(None, 4, None, _K1_PAD)
            need_dx = ctx.needs_input_grad[0]
            if need_dx:
                # [M, x_cols]: when the caller passed the zero-padded input, the pad columns of its
                # gradient are never read (they belong to a constant) and stay unwritten
                dx = f32(M, ctx.x_cols)
                data_grad(4, w5[:, :cin], dx, False, None)
            dWy, db5, nparts = layer_bwd(4, 3, W=w5[:, cin:])
            Gs[4] = None
            for i in (3, 2, 1):
                finish(i, nparts)
                dW, db, nparts = layer_bwd(i, i - 1)
                g_conv_w[i], g_conv_b[i] = dW.unsqueeze(-1), db
                Gs[i] = None
            finish(0, nparts)
            dW0, db0 = wgrad(None, 0, None, _K1_PAD)
            if need_dx:
                data_grad(0, conv_w[0], dx, True, None)
        if njobs[0]:
            if side is not None:
                main_stream.wait_stream(side)
            _native.ganet_check(lib.ganet_wgrad_reduce_batch(njobs[0], jobs, st))
        if heads:
            g_conv_w[4], g_conv_b[4] = torch.cat([dWx[:, :cin], dWy], 1).unsqueeze(-1), db5
            g_conv_w[0], g_conv_b[0] = dW0[:, :cin].contiguous().unsqueeze(-1), db0
        grads = []
        for i in range(nl):
            grads += [g_conv_w[i], g_conv_b[i], g_gamma[i], g_beta[i]]
        for j in range(3):
            grads += [g_out_w[j], g_out_b[j]]
        return (dx, None, None) + tuple(grads)


def _decoder_grad_views(params, nl, dev):
    """GanetDecoderGrads over views of ONE buffer (a single allocation instead of ~50) shaped like the parameters."""
    from ._native import GanetDecoderGrads
    shapes = [tuple(p.shape) for p in params]
    sizes = [(p.numel() + 3) // 4 * 4 for p in params]            # 16-byte aligned pieces
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    views, off = [], 0
    for shp, n, p in zip(shapes, sizes, params):
        views.append(flat[off:off + p.numel()].view(shp))
        off += n
    G = GanetDecoderGrads()
    for i in range(nl):
        G.dW[i], G.db[i], G.dgamma[i], G.dbeta[i] = (views[4 * i + k].data_ptr() for k in range(4))
    for j in range(3):
        G.dW8[j], G.db8[j] = views[4 * nl + 2 * j].data_ptr(), views[4 * nl + 2 * j + 1].data_ptr()
    return G, views


def _decoder_head_grads(d_outs, params, nl, M, dev):
    """The three heads' output gradients as contiguous tensors (a head the objective does not use: zeros) + pointers."""
    widths = [int(p.shape[0]) for p in params[4 * nl::2]]          # conv8 / conv8N / conv8SH: 3, 1, 3 columns
    douts = [d.contiguous() if d is not None else torch.zeros((M, w), dtype=torch.float32, device=dev)
             for d, w in zip(d_outs, widths)]
    return douts, (ctypes.c_void_p * 3)(*[d.data_ptr() for d in douts])


def _decoder_bwd_native(ctx, lib, sv, d_outs):
    nl, cin, dec = ctx.nl, ctx.cin, ctx.dec
    xp, saved, params = sv[0], sv[1], sv[2:]
    dev, M = xp.device, xp.shape[0]
    P, keep = _native_decoder_params(dec, params, nl)
    G, views = _decoder_grad_views(params, nl, dev)
    dx = None
    if ctx.needs_input_grad[0]:
        # [M, x_cols]: the pad columns of a zero-padded input belong to a constant and stay unwritten
        dx = torch.empty((M, ctx.x_cols), dtype=torch.float32, device=dev)
        G.dx, G.x_cols = dx.data_ptr(), ctx.x_cols
    douts, dptrs = _decoder_head_grads(d_outs, params, nl, M, dev)
    wsb = lib.ganet_decoder_bwd_workspace(M)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    side = _side_stream(dev) if (_WGRAD_STREAM and not _profiling) else None
    # (the side stream's launches are ordered against the main stream inside the call, events both ways: every buffer is
    # free again in main-stream order, no record_stream needed — it would only delay the reuse of the workspace)
    _native.ganet_check(lib.ganet_decoder_bwd(M, _ptr(xp), ctypes.byref(P), _ptr(saved), dptrs, ctypes.byref(G), _ptr(ws),
                                              wsb, _stream(dev), None if side is None else ctypes.c_void_p(side.cuda_stream)))
    return (dx, None, None) + tuple(views)


_DecoderFn._backward_native = staticmethod(_decoder_bwd_native)


def _allreduce_partials(col_parts, width: int) -> None:
    """col_parts: list of per-workgroup partial column sums [nparts_k, width] (layers of one level) -> every buffer
    holds the sums over ALL ranks' partials in its first row and zeros elsewhere (the statistics kernels add the rows
    up). ONE all-reduce for the whole list; the local reductions and the all-reduce run in float64."""
    parts = [cp.view(-1, width) for cp in col_parts]
    if len({p.shape for p in parts}) == 1:
        tot = torch.stack(parts).double().sum(1)                 # [k, width] in one reduction launch
    else:
        tot = torch.stack([p.double().sum(0) for p in parts])
    parallel.all_reduce_sum_(tot)
    totf = tot.float()
    for k, p in enumerate(parts):
        p.zero_()
        p[0] = totf[k]


def decoder_mlp(dec, x, m_global=None):
    """ShapeDecoder.forward_points on the fused kernels: x [M, in_size] -> (residual [M,3],
    scale logits [M,1], colour logits [M,3]) — the sigmoids of the two heads stay with the caller.
    m_global: row count of the global batch when x holds only this rank's rows (see _DecoderFn)."""
    return _DecoderFn.apply(x, dec, m_global, *_decoder_param_list(dec))


def _decoder_param_list(dec):
    pairs, outs, _sig = _decoder_modules(dec)
    params = []
    for c, b in pairs:
        params += [c.weight, c.bias, b.weight, b.bias]
    for c in outs:
        params += [c.weight, c.bias]
    return params


# ------------------------------------------------------------------------------------------------
# The stage-2 pose encoder (network.UnetNoCond5DS) as one native call each way (csrc/ganet_unet.hip).
_UNET_BN = ("conv2", "conv3", "conv4", "upconv1", "upconv2", "upconv3", "upconv4")
_NATIVE_UNET = _dev.knobs.native_unet


def unet_supported(net, x) -> bool:
    """net: network.UnetNoCond5DS; x [B, cin, S, S]. Mirrors `unet_ok` of csrc/ganet_unet.hip (S a power of two >= 32,
    B > 0, cin <= 8, nf and cout multiples of 32); the native kernels produce no input gradient, so an input that
    requires one takes the torch path."""
    if not (_NATIVE_UNET and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] == x.shape[3]):
        return False
    S = int(x.shape[2])
    if x.shape[0] < 1 or S < 32 or (S & (S - 1)) or (x.requires_grad and torch.is_grad_enabled()):
        return False
    ups = [getattr(net, f"upconv{k}") for k in range(1, 6)]
    if not all(isinstance(u.up, torch.nn.ConvTranspose2d) and not u.use_dropout for u in ups):
        return False
    nf, cout = net.conv1.conv.out_channels, net.upconv5.up.out_channels
    bns = [getattr(net, n).bn for n in _UNET_BN]
    plain = all(type(b) is torch.nn.BatchNorm2d and not b.affine and b.momentum is not None and b.track_running_stats
                and b.training == net.training for b in bns)
    return (plain and nf % 32 == 0 and cout % 32 == 0 and x.shape[1] <= 8
            and net.conv1.conv.in_channels == x.shape[1] and net.upconv5.up.bias is not None
            and (net.training or not torch.is_grad_enabled()))


def _unet_params(net, ws, S):
    P = _native.GanetUnetParams()
    P.cin, P.nf, P.cout, P.S = net.conv1.conv.in_channels, net.conv1.conv.out_channels, net.upconv5.up.out_channels, S
    for k in range(5):
        P.Wd[k], P.Wu[k] = ws[k].data_ptr(), ws[5 + k].data_ptr()
    P.bias5 = ws[10].data_ptr()
    for i, name in enumerate(_UNET_BN):
        bn = getattr(net, name).bn
        P.running_mean[i], P.running_var[i] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        P.num_batches_tracked[i] = bn.num_batches_tracked.data_ptr()
    bn0 = getattr(net, _UNET_BN[0]).bn
    P.eps, P.momentum = float(bn0.eps), float(bn0.momentum)
    return P


class _UnetFn(torch.autograd.Function):
    """(x [B,cin,S,S], conv1..5 weights, upconv1..5 weights, upconv5 bias) -> pose features, a logical-NCHW view of a
    channels-last buffer [B,S,S,cout]."""

    @staticmethod
    def forward(ctx, x, net, *ws):
        lib = _native.ganet()
        dev = x.device
        B, _, S, _ = x.shape
        xc = x.contiguous()
        ws = tuple(w if w.is_contiguous() else w.contiguous() for w in ws)
        P = _unet_params(net, ws, S)
        saved = torch.empty(lib.ganet_unet_saved_floats(ctypes.byref(P), B), dtype=torch.float32, device=dev)
        out = torch.empty((B, S, S, P.cout), dtype=torch.float32, device=dev)
        wsb = lib.ganet_unet_fwd_workspace(ctypes.byref(P), B)
        wk = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _native.ganet_check(lib.ganet_unet_fwd(ctypes.byref(P), B, _ptr(xc), int(net.training), _ptr(saved), _ptr(out),
                                               _ptr(wk), wsb, _stream(dev)))
        ctx.net = net
        ctx.save_for_backward(xc, saved, *ws)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _native.ganet()
        xc, saved, *ws = ctx.saved_tensors
        dev = xc.device
        B, _, S, _ = xc.shape
        P = _unet_params(ctx.net, ws, S)
        gh = g.permute(0, 2, 3, 1).contiguous()                        # channels-last (no copy if it already is)
        grads = [torch.empty_like(w) for w in ws]
        G = _native.GanetUnetGrads()
        for k in range(5):
            G.dWd[k], G.dWu[k] = grads[k].data_ptr(), grads[5 + k].data_ptr()
        G.dbias5 = grads[10].data_ptr()
        wsb = lib.ganet_unet_bwd_workspace(ctypes.byref(P), B)
        wk = torch.empty(wsb, dtype=torch.uint8, device=dev)
        # (weight gradients on the side stream, ordered inside the call with events both ways: see ganet_decoder_bwd)
        side = _side_stream(dev) if (_dev.knobs.unet_wgrad_stream and not _profiling) else None
        _native.ganet_check(lib.ganet_unet_bwd(ctypes.byref(P), B, _ptr(xc), _ptr(saved), _ptr(gh), ctypes.byref(G), _ptr(wk),
                                               wsb, _stream(dev), None if side is None else ctypes.c_void_p(side.cuda_stream)))
        return (None, None) + tuple(grads)


def unet_forward(net, x):
    ws = [getattr(net, f"conv{k}").conv.weight for k in range(1, 6)]
    ws += [getattr(net, f"upconv{k}").up.weight for k in range(1, 6)]
    ws.append(net.upconv5.up.bias)
    return _UnetFn.apply(x, net, *ws)
