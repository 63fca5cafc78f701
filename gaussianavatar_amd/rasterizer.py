"""Python surface of the rasterizer: a mirror of the `diff_gaussian_rasterization` package.

The reference imports `GaussianRasterizationSettings` and `GaussianRasterizer` from that
(un-vendored, CUDA) package at /root/reference/gaussian_renderer/__init__.py:6 and calls them
at :21-48. This module provides the same names, argument meaning, return values and error
behaviour on top of the gfx950 HIP library (include/gsr.h); the top-level package
`diff_gaussian_rasterization/` re-exports it so the reference's renderer shim runs unchanged.

Everything is enqueued on torch's current HIP stream; the forward pass performs no
device->host read-back (the reference blocks on a cudaMemcpy of the pair count every frame).
The only thing the host ever needs to learn — whether the (tile,Gaussian) pair buffer was
large enough — comes back through an asynchronous copy into pinned memory that is polled
later (see `_PairCapacity`).
"""
from __future__ import annotations

import ctypes
import threading
import time
import warnings
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _native


class GaussianRasterizationSettings(NamedTuple):
    """Same 12 fields, same order as the reference builds them
    (/root/reference/gaussian_renderer/__init__.py:21-34)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class RasterizerOverflow(RuntimeError):
    """The (tile, Gaussian) pair buffer of an earlier forward pass was too small: that frame
    was rendered from truncated tile lists. The capacity has been raised for later calls."""


class _PairCapacity:
    """Sizing of the (tile,Gaussian) pair buffer without a host sync on the steady-state path.

    Per problem shape (P, W, H):
      * the FIRST forward is checked synchronously and transparently re-rendered with a larger
        buffer if it overflowed (one sync, once);
      * afterwards capacity = max(pairs_per_gaussian * P, floor, 2 x the largest count seen),
        the 8 status words are copied asynchronously into pinned memory after every forward and
        polled (never waited for) on later calls — so the buffer tracks the scene with a 2x
        margin and only a frame-to-frame doubling of the pair count can overflow it;
      * an overflow that still happens yields NO gradient: the backward pass of that frame writes zeros and
        raises a device-side flag (`overflow_flag(device)`) that `gaussianavatar_amd.optim.Adam` reads to skip
        its step — all without a host sync, so a step computed from truncated tile lists is never applied
        (upstream resizes and never truncates; here the truncated step is dropped and the buffer is resized for
        the following calls). The event itself is detected on a later poll: by default it WARNS
        (`overflow_events` counts them — a long run is not aborted by one transient spike); policy "raise"
        turns it into RasterizerOverflow for callers that would rather stop;
      * forwards that nobody can differentiate (evaluation) and settings.debug are always
        checked synchronously and re-rendered when needed.
    """

    def __init__(self):
        self.pairs_per_gaussian = 16
        self.floor = 1 << 16
        self.seen = {}         # (P, W, H) -> largest pair count seen
        self.stamp = {}        # (P, W, H) -> time of the last status seen
        self.policy = "warn"
        self.overflow_events = 0
        self.pending = []      # (event, pinned status, capacity, key)
        self.pool = []
        self.last_status = None
        self.pairs_sum = 0       # running statistics over checked forward passes
        self.calls = 0

    STALE_SECONDS = 2.0

    def known(self, key) -> bool:
        """History exists and is fresh (training calls arrive every few ms; after a pause the
        next forward is re-checked synchronously, which costs one sync, once)."""
        return key in self.seen and (time.monotonic() - self.stamp.get(key, 0.0)) < self.STALE_SECONDS

    def reset(self):
        self.poll(block=True)
        self.seen.clear()
        self.stamp.clear()

    def capacity(self, key) -> int:
        cap = max(self.pairs_per_gaussian * key[0], self.floor, 2 * self.seen.get(key, 0) + 1024)
        return min(cap, 0xfffffff0)

    def _pinned(self):
        return self.pool.pop() if self.pool else torch.empty(8, dtype=torch.int32).pin_memory()

    def post(self, status_dev: torch.Tensor, cap: int, key):
        host = self._pinned()
        host.copy_(status_dev, non_blocking=True)
        self.posted(host, cap, key)

    def posted(self, host: torch.Tensor, cap: int, key):
        """`host` (from _pinned()) is being written by work already queued on the current stream — by a copy (post)
        or by the status kernel itself (pinned host memory is device-addressable: the batched path saves the copy)."""
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, host, cap, key))

    def _account(self, host, key, cap=None):
        needed, overflow = int(host[0]), int(host[1])
        self.last_status = host.tolist()
        self.pool.append(host)
        # (the segment records the forward pass leaves for the backward pass cannot overflow: their slots are an
        # exact function of the tile lists, include/gsr.h GsrLayout.seg_entries — the pair buffer is the only capacity)
        self.seen[key] = max(self.seen.get(key, 0), needed)
        if overflow and cap is not None:
            self.seen[key] = max(self.seen[key], int(cap))
        self.stamp[key] = time.monotonic()
        frames = int(host[5]) if int(host[5]) > 0 else 1          # batched launches report totals
        self.pairs_sum += int(host[4]) if int(host[5]) > 0 else needed
        self.calls += frames
        return needed, overflow

    def poll(self, block: bool = False):
        """Check finished status copies; with block=True wait for all of them."""
        while self.pending:
            ev, host, cap, key = self.pending[0]
            if not block and len(self.pending) <= 8 and not ev.query():
                break
            ev.synchronize()
            self.pending.pop(0)
            needed, overflow = self._account(host, key, cap)
            if overflow:
                self.overflow_events += 1
                msg = (f"rasterizer pair buffer overflow: a forward pass needed {needed} "
                       f"(tile,Gaussian) pairs but had room for {cap}; that frame was rendered "
                       f"from truncated tile lists. Capacity is now raised; to avoid this up front "
                       f"call gaussianavatar_amd.rasterizer.set_pair_capacity(pairs_per_gaussian=...).")
                if self.policy == "raise":
                    raise RasterizerOverflow(msg)
                warnings.warn(msg)

    def wait_last(self):
        """Synchronously resolve the newest pending copy WITHOUT raising; returns (needed, overflow)."""
        ev, host, cap, key = self.pending.pop()
        ev.synchronize()
        return self._account(host, key, cap)


_capacity = _PairCapacity()


def set_pair_capacity(pairs_per_gaussian: Optional[int] = None, floor: Optional[int] = None,
                      on_overflow: Optional[str] = None) -> None:
    """Tune the pair-buffer sizing policy (see _PairCapacity)."""
    if pairs_per_gaussian is not None:
        _capacity.pairs_per_gaussian = int(pairs_per_gaussian)
    if floor is not None:
        _capacity.floor = int(floor)
    if on_overflow is not None:
        assert on_overflow in ("raise", "warn")
        _capacity.policy = on_overflow


def reset_capacity_history() -> None:
    """Forget the pair counts seen so far (call when the scene changes abruptly, e.g. a new
    model): the next forward of every shape is checked synchronously again."""
    _capacity.reset()


def check_overflow(block: bool = True) -> None:
    """Resolve outstanding overflow checks (block=True waits for the device)."""
    _capacity.poll(block=block)


def overflow_events() -> int:
    """How many steady-state forward passes were rendered from truncated tile lists so far."""
    return _capacity.overflow_events


def pair_statistics(reset: bool = False):
    """(forward passes checked, mean (tile,Gaussian) pairs per pass) since the last reset."""
    check_overflow(block=True)
    n, s = _capacity.calls, _capacity.pairs_sum
    if reset:
        _capacity.calls, _capacity.pairs_sum = 0, 0
    return n, (s / n if n else 0.0)


_overflow_flags = {}


def overflow_flag(device) -> torch.Tensor:
    """The device-side int32[1] flag of `device` that a backward pass raises when its forward pass had overflowed
    (that frame's gradients are zeros, include/gsr.h: gsr_backward). `optim.Adam.step` skips its update while the
    flag is set and lowers it behind itself; `optim.Adam.zero_grad` (hence AvatarModel.zero_grad) lowers it too, so the
    flag always describes the backward passes since the last zero_grad / step."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    f = _overflow_flags.get(idx)
    if f is None:
        f = _overflow_flags[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return f


def clear_overflow_flag(device=None) -> None:
    """Lower the flag(s) by hand (one 4-byte fill) — loops that use neither optim.Adam.step nor its zero_grad."""
    for idx, f in _overflow_flags.items():
        if device is None or torch.device(device).index in (None, idx):
            f.zero_()


PROFILE_KERNELS = ("preprocess", "tile_scan", "scatter", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd")
_profile = None       # the GsrProfile object of this module (caller-owned, bound to the launching threads)
_profile_mask = 0
_profile_tls = threading.local()


def _profile_bind_this_thread() -> None:
    """The binding is per thread (include/gsr.h) and autograd runs backward on its own thread: every entry into the
    library re-binds when the requested mask differs from what this thread last bound (one integer compare)."""
    if getattr(_profile_tls, "mask", 0) != _profile_mask and _profile is not None:
        _native.gsr_check(_native.gsr().gsr_profile_bind(_profile, _profile_mask))
        _profile_tls.mask = _profile_mask


def profile_enable(on=True) -> None:
    """Bracket rasterizer kernel launches of the calling thread with HIP events on their stream (bench only).
    `on`: True = every kernel, False = off, or an iterable of kernel names (PROFILE_KERNELS) to time only those."""
    global _profile, _profile_mask
    lib = _native.gsr()
    if on is True:
        mask = (1 << len(PROFILE_KERNELS)) - 1
    elif not on:
        mask = 0
    else:
        mask = sum(1 << PROFILE_KERNELS.index(k) for k in on)
    if _profile is None:
        if not mask:
            return
        _profile = ctypes.c_void_p(lib.gsr_profile_create())
    _profile_mask = mask
    _profile_bind_this_thread()


def profile_read(reset: bool = True) -> dict:
    """{kernel name: (total ms, launches)} measured by the in-library HIP events. Blocks until
    the recorded events have completed."""
    lib = _native.gsr()
    ms = (ctypes.c_double * 7)()
    n = (ctypes.c_int64 * 7)()
    if _profile is not None:
        _native.gsr_check(lib.gsr_profile_read(_profile, ms, n, 1 if reset else 0))
    return {lib.gsr_profile_kernel_name(i).decode(): (ms[i], int(n[i])) for i in range(7)}


def last_status():
    """The 8 status words of the most recently checked forward pass:
    [pairs needed, overflow flag, -, longest tile list, ...]."""
    return _capacity.last_status


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, shape=None) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    if shape is not None and tuple(t.shape) != tuple(shape):
        t = t.reshape(shape)
    return t.contiguous()


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(_native.raw_stream(device))


def _settings_struct(rs: GaussianRasterizationSettings, keep: list) -> "_native.GsrSettings":
    dev = rs.viewmatrix.device
    bg = _f32c(rs.bg.to(dev), (3,))
    view = _f32c(rs.viewmatrix, (16,))
    proj = _f32c(rs.projmatrix.to(dev), (16,))
    campos = _f32c(rs.campos.to(dev), (3,))
    keep += [bg, view, proj, campos]
    return _native.GsrSettings(int(rs.image_height), int(rs.image_width), float(rs.tanfovx),
                               float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree),
                               int(bool(rs.prefiltered)), int(bool(rs.debug)),
                               bg.data_ptr(), view.data_ptr(), proj.data_ptr(), campos.data_ptr())


def workspace_views(workspace: torch.Tensor, P: int, W: int, H: int, max_pairs: int) -> dict:
    """Typed views of the published sub-arrays of a forward workspace (for tests/tools)."""
    lib = _native.gsr()
    L = _native.GsrLayout()
    _native.gsr_check(lib.gsr_workspace_layout(P, W, H, max_pairs, ctypes.byref(L)))
    T = ((W + 15) // 16) * ((H + 15) // 16)
    edge = lib.gsr_render_block_edge()           # pixel block of a wave: 8 -> 4 blocks per tile
    Bk, npx = (16 // edge) ** 2, edge * edge
    S = max_pairs * Bk // 64 + Bk * T + Bk       # segment slots (gsr_common.h: seg_capacity)

    def view(off, nbytes, dtype, shape):
        return workspace[off:off + nbytes].view(dtype).reshape(shape)

    return dict(
        depth=view(L.depth, P * 4, torch.float32, (P,)),
        xy=view(L.xy, P * 8, torch.float32, (P, 2)),
        conic_opacity=view(L.conic_opacity, P * 16, torch.float32, (P, 4)),
        rgb=view(L.rgb, P * 16, torch.float32, (P, 4)),
        cov3d=view(L.cov3d, P * 24, torch.float32, (P, 6)),
        rect=view(L.rect, P * 16, torch.int32, (P, 4)),
        tiles_touched=view(L.tiles_touched, P * 4, torch.int32, (P,)),
        tile_count=view(L.tile_count, T * 4, torch.int32, (T,)),
        tile_offset=view(L.tile_offset, (T + 1) * 4, torch.int32, (T + 1,)),
        point_list=view(L.point_list, max_pairs * 4, torch.int32, (max_pairs,)),
        final_T=view(L.final_T, W * H * 4, torch.float32, (H * W,)),
        n_contrib=view(L.n_contrib, W * H * 4, torch.int32, (H * W,)),
        grad_acc=view(L.grad_acc, P * 64, torch.float32, (P, 16)),
        status=view(L.status, 32, torch.int32, (8,)),
        seg_entries=view(L.seg_entries, S * 512, torch.int32, (S, 64, 2)),
        seg_ckpt=view(L.seg_ckpt, S * npx * 16, torch.float32, (S, npx, 4)),
        seg_info=view(L.seg_info, S * 8, torch.int32, (S, 2)),
        pix_accum=view(L.pix_accum, W * H * 16, torch.float32, (H * W, 4)),
        seg_count=view(L.seg_count, T * Bk * 4, torch.int32, (T, Bk)),
    )


def _ws_mode(rs, record: bool) -> int:
    """Workspace mode of a call (include/gsr.h GSR_WS_*): forward-only renders use the small workspace."""
    if not record:
        return _native.GSR_WS_EVAL
    return _native.GSR_WS_DEBUG if bool(rs.debug) else _native.GSR_WS_TRAIN


def _forward_once(rs, means3D, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                  max_pairs, shs=None, record=True):
    lib = _native.gsr()
    _profile_bind_this_thread()
    dev = means3D.device
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    nbytes = lib.gsr_workspace_bytes_for(P, W, H, max_pairs, _ws_mode(rs, record))
    if nbytes == 0:
        raise RuntimeError("gsr_workspace_bytes_for: invalid arguments")
    workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    keep = []
    st = _settings_struct(rs, keep)
    fwd = lib.gsr_forward if record else lib.gsr_forward_eval
    rc = fwd(ctypes.byref(st), P, _ptr(means3D), _ptr(colors_precomp), _ptr(shs),
             0 if shs is None else int(shs.shape[1]), _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
             _ptr(workspace), nbytes, max_pairs, _ptr(color), _ptr(radii), _stream_ptr(dev))
    _native.gsr_check(rc)
    L = _native.GsrLayout()
    lib.gsr_workspace_layout(P, W, H, max_pairs, ctypes.byref(L))
    status = workspace[L.status:L.status + 32].view(torch.int32)      # (the offsets do not depend on the mode)
    return color, radii, workspace, status


def rasterize_with_state(raster_settings, means3D, colors_precomp, opacities, scales=None,
                         rotations=None, cov3D_precomp=None, max_pairs: Optional[int] = None):
    """Forward only, returning the internal state as well (parity tests and tools):
    (color, radii, views, status_list) where `views` are the typed workspace sub-arrays."""
    P = means3D.shape[0]
    means3D = _f32c(means3D, (P, 3))
    colors_precomp = _f32c(colors_precomp, (P, 3))
    opacities = _f32c(opacities, (P,))
    scales = _f32c(scales, (P, 3)) if scales is not None else None
    rotations = _f32c(rotations, (P, 4)) if rotations is not None else None
    cov3D_precomp = _f32c(cov3D_precomp, (P, 6)) if cov3D_precomp is not None else None
    if max_pairs is None:
        max_pairs = _capacity.capacity((P, int(raster_settings.image_width), int(raster_settings.image_height)))
    color, radii, workspace, status = _forward_once(
        raster_settings, means3D, colors_precomp, opacities, scales, rotations, cov3D_precomp, max_pairs)
    views = workspace_views(workspace, P, int(raster_settings.image_width),
                            int(raster_settings.image_height), max_pairs)
    torch.cuda.synchronize()
    return color, radii, views, status.tolist()


class _RasterizeGaussians(torch.autograd.Function):
    """Argument order and gradient order follow the upstream autograd Function
    (SURVEY.md §8b): (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
    cov3Ds_precomp, raster_settings)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings, sync_check=False):
        rs = raster_settings
        if not means3D.is_cuda:
            raise RuntimeError("GaussianRasterizer: tensors must live on a HIP device "
                               "(there is no CPU fallback)")
        P = means3D.shape[0]
        means3D = _f32c(means3D, (P, 3))
        if sh is not None and sh.numel() == 0:
            sh = None
        if colors_precomp is not None and colors_precomp.numel() == 0 and sh is not None:
            colors_precomp = None
        if sh is not None:                      # [P, M, 3], M >= (sh_degree+1)^2
            sh = _f32c(sh, (P, sh.shape[-2], 3))
        else:
            colors_precomp = _f32c(colors_precomp, (P, 3))
        opacities = _f32c(opacities, (P,))
        scales = _f32c(scales, (P, 3)) if scales is not None else None
        rotations = _f32c(rotations, (P, 4)) if rotations is not None else None
        cov3Ds_precomp = _f32c(cov3Ds_precomp, (P, 6)) if cov3Ds_precomp is not None else None

        _capacity.poll()
        key = (P, int(rs.image_width), int(rs.image_height))
        # sync_check is set by rasterize_gaussians when nobody can call backward (evaluation, torch.no_grad()): such a
        # render records nothing for a backward pass (gsr_forward_eval: no segment records, the small workspace)
        record = not bool(sync_check)
        sync_check = bool(rs.debug) or bool(sync_check) or not _capacity.known(key)
        max_pairs = _capacity.capacity(key)
        while True:
            color, radii, workspace, status = _forward_once(
                rs, means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, max_pairs, sh, record)
            _capacity.post(status, max_pairs, key)
            if not sync_check:
                break
            needed, overflow = _capacity.wait_last()
            if not overflow:
                break
            max_pairs = _capacity.capacity(key)

        ctx.raster_settings = rs
        ctx.max_pairs = max_pairs
        ctx.has_sr = scales is not None
        ctx.has_sh = sh is not None
        ctx.recorded = record
        ctx.save_for_backward(means3D, sh if sh is not None else colors_precomp, opacities,
                              scales if scales is not None else torch.empty(0, device=means3D.device),
                              rotations if rotations is not None else torch.empty(0, device=means3D.device),
                              cov3Ds_precomp if cov3Ds_precomp is not None else torch.empty(0, device=means3D.device),
                              radii, workspace)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)         # (no [P] zero tensor for radii's "gradient")
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _grad_radii):
        lib = _native.gsr()
        rs = ctx.raster_settings
        means3D, colors_precomp, opacities, scales, rotations, cov3D, radii, workspace = ctx.saved_tensors
        if not ctx.recorded:
            raise RuntimeError("GaussianRasterizer: this render was made without gradient tracking (forward-only path)")
        if grad_color is None:
            return (None,) * 10
        if not ctx.has_sr:
            scales = rotations = None
        else:
            cov3D = None
        _capacity.poll()
        _profile_bind_this_thread()
        dev = means3D.device
        P = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        grad_color = _f32c(grad_color, (3, H, W))
        need = ctx.needs_input_grad

        def out(flag, *shape):
            return torch.empty(shape, dtype=torch.float32, device=dev) if flag else None

        d_means3D = out(need[0], P, 3)
        d_means2D = out(need[1], P, 3)
        sh = None
        if ctx.has_sh:
            sh, colors_precomp = colors_precomp, None
        d_sh = out(need[2] and ctx.has_sh, P, sh.shape[1] if ctx.has_sh else 0, 3)
        d_colors = out(need[3] and not ctx.has_sh, P, 3)
        d_opac = out(need[4], P, 1)
        d_scales = out(need[5] and ctx.has_sr, P, 3)
        d_rots = out(need[6] and ctx.has_sr, P, 4)
        d_cov = out(need[7] and not ctx.has_sr, P, 6)
        keep = []
        st = _settings_struct(rs, keep)
        rc = lib.gsr_backward(ctypes.byref(st), P, _ptr(means3D), _ptr(colors_precomp), _ptr(sh),
                              int(sh.shape[1]) if ctx.has_sh else 0, _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3D),
                              _ptr(radii), _ptr(workspace), workspace.numel(), ctx.max_pairs,
                              _ptr(grad_color), _ptr(d_means3D), _ptr(d_means2D), _ptr(d_colors),
                              _ptr(d_sh), _ptr(d_opac), _ptr(d_scales), _ptr(d_rots), _ptr(d_cov),
                              _ptr(overflow_flag(dev)), _stream_ptr(dev))
        _native.gsr_check(rc)
        return d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rots, d_cov, None, None


def _frame_stride(t: torch.Tensor, frames: int, inner_shape) -> tuple:
    """(tensor whose data pointer is frame 0, element stride between frames). A leading dimension
    of size 1 or an expanded (stride 0) one means "shared by all frames"."""
    inner = 1
    for d in inner_shape:
        inner *= d
    if t.dim() == len(inner_shape):                         # no frame dimension at all
        return _f32c(t, inner_shape), 0
    if t.shape[0] == 1 or t.stride(0) == 0:
        return _f32c(t[0], inner_shape), 0
    assert t.shape[0] == frames, (t.shape, frames)
    return _f32c(t, (frames,) + tuple(inner_shape)), inner


class _RasterizeGaussiansBatch(torch.autograd.Function):
    """All frames of a batch through ONE launch of every rasterizer kernel (gsr_forward_batch).
    Inputs: means3D [B,P,3]; colors_precomp, scales [B,P,3] (or shared [P,3] / expanded views);
    opacities [P,1] | [B,P,1]; rotations [P,4] | [B,P,4]; settings.viewmatrix / projmatrix
    [B,4,4] (or shared [4,4]). Returns (color [B,3,H,W], radii [B,P])."""

    @staticmethod
    def forward(ctx, means3D, colors_precomp, opacities, scales, rotations, raster_settings, sync_check):
        rs = raster_settings
        if not means3D.is_cuda:
            raise RuntimeError("GaussianRasterizer: tensors must live on a HIP device "
                               "(there is no CPU fallback)")
        lib = _native.gsr()
        _profile_bind_this_thread()
        dev = means3D.device
        B, P = means3D.shape[0], means3D.shape[1]
        H, W = int(rs.image_height), int(rs.image_width)
        means3D = _f32c(means3D, (B, P, 3))
        col, s_col = _frame_stride(colors_precomp, B, (P, 3))
        opa_flat = opacities.squeeze(-1) if (opacities.dim() >= 2 and opacities.shape[-1] == 1) else opacities
        opa, s_opa = _frame_stride(opa_flat, B, (P,))
        sca, s_sca = _frame_stride(scales, B, (P, 3))
        rot, s_rot = _frame_stride(rotations, B, (P, 4))
        view, s_view = _frame_stride(rs.viewmatrix.reshape(*rs.viewmatrix.shape[:-2], 16), B, (16,))
        proj, s_proj = _frame_stride(rs.projmatrix.reshape(*rs.projmatrix.shape[:-2], 16), B, (16,))
        bg = _f32c(rs.bg.to(dev), (3,))
        campos = _f32c(rs.campos.to(dev).reshape(-1)[:3], (3,))
        st = _native.GsrSettings(H, W, float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
                                 int(rs.sh_degree), int(bool(rs.prefiltered)), int(bool(rs.debug)),
                                 bg.data_ptr(), view.data_ptr(), proj.data_ptr(), campos.data_ptr())
        bt = _native.GsrBatch(B, P * 3, s_col, s_opa, s_sca, s_rot, 0, s_view, s_proj, 0, 0)
        _capacity.poll()
        key = (P, W, H)
        record = not bool(sync_check)            # nobody can call backward: forward-only kernels, small workspace
        sync_check = bool(rs.debug) or bool(sync_check) or not _capacity.known(key)
        max_pairs = _capacity.capacity(key)
        mode = _ws_mode(rs, record)
        fwd = lib.gsr_forward_batch if record else lib.gsr_forward_eval_batch
        while True:
            frame_bytes = lib.gsr_workspace_bytes_for(P, W, H, max_pairs, mode)
            workspace = torch.empty(B * frame_bytes, dtype=torch.uint8, device=dev)
            color = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((B, P), dtype=torch.int32, device=dev)
            _native.gsr_check(fwd(
                ctypes.byref(st), ctypes.byref(bt), P, _ptr(means3D), _ptr(col), None, 0, _ptr(opa),
                _ptr(sca), _ptr(rot), None, _ptr(workspace), workspace.numel(), max_pairs,
                _ptr(color), _ptr(radii), _stream_ptr(dev)))
            # one record for the whole launch: [max pairs of a frame, any overflow, -, longest tile
            # list, total pairs of all frames, frames, -, -] (gsr_batch_status, one tiny kernel)
            worst = _capacity._pinned()          # written by the kernel over the host link: no device-to-host copy
            _native.gsr_check(lib.gsr_batch_status(_ptr(workspace), B, P, W, H, max_pairs, mode, _ptr(worst),
                                                   _stream_ptr(dev)))
            _capacity.posted(worst, max_pairs, key)
            if not sync_check:
                break
            needed, overflow = _capacity.wait_last()
            if not overflow:
                break
            max_pairs = _capacity.capacity(key)
        ctx.raster_settings = rs
        ctx.max_pairs = max_pairs
        ctx.recorded = record
        # (original shape, has a leading frame dimension) of every differentiable input
        ctx.meta = (B, P, H, W, (s_col, s_opa, s_sca, s_rot, s_view, s_proj),
                    (tuple(colors_precomp.shape), colors_precomp.dim() == 3),
                    (tuple(opacities.shape), opa_flat.dim() == 2),
                    (tuple(scales.shape), scales.dim() == 3),
                    (tuple(rotations.shape), rotations.dim() == 3))
        ctx.save_for_backward(means3D, col, opa, sca, rot, view, proj, bg, campos, radii, workspace)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)         # (autograd would otherwise fill a [B,P] zero tensor for radii's "gradient")
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _grad_radii):
        lib = _native.gsr()
        rs = ctx.raster_settings
        means3D, col, opa, sca, rot, view, proj, bg, campos, radii, workspace = ctx.saved_tensors
        B, P, H, W, strides, col_shape, opa_shape, sca_shape, rot_shape = ctx.meta
        if not ctx.recorded:
            raise RuntimeError("rasterize_gaussians_batch: this render was made without gradient tracking")
        if grad_color is None:
            return (None,) * 7
        s_col, s_opa, s_sca, s_rot, s_view, s_proj = strides
        dev = means3D.device
        _capacity.poll()
        _profile_bind_this_thread()
        grad_color = _f32c(grad_color, (B, 3, H, W))
        need = ctx.needs_input_grad

        def out(flag, *shape):
            return torch.empty(shape, dtype=torch.float32, device=dev) if flag else None

        d_means = out(need[0], B, P, 3)
        d_col = out(need[1], B, P, 3)
        d_opa = out(need[2], B, P)
        d_sca = out(need[3], B, P, 3)
        d_rot = out(need[4], B, P, 4)
        st = _native.GsrSettings(H, W, float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
                                 int(rs.sh_degree), int(bool(rs.prefiltered)), int(bool(rs.debug)),
                                 bg.data_ptr(), view.data_ptr(), proj.data_ptr(), campos.data_ptr())
        bt = _native.GsrBatch(B, P * 3, s_col, s_opa, s_sca, s_rot, 0, s_view, s_proj, 0, 0)
        _native.gsr_check(lib.gsr_backward_batch(
            ctypes.byref(st), ctypes.byref(bt), P, _ptr(means3D), _ptr(col), None, 0, _ptr(opa), _ptr(sca),
            _ptr(rot), None, _ptr(radii), _ptr(workspace), workspace.numel(), ctx.max_pairs,
            _ptr(grad_color), _ptr(d_means), None, _ptr(d_col), None, _ptr(d_opa), _ptr(d_sca), _ptr(d_rot),
            None, _ptr(overflow_flag(dev)), _stream_ptr(dev)))

        def fold(g, info):
            """Per-frame gradients [B, ...] back to the shape the caller passed: summed over frames
            for inputs without a frame dimension or with a broadcast one of size 1."""
            if g is None:
                return None
            shape, has_frame_dim = info
            if not has_frame_dim or (shape[0] == 1 and B > 1):
                g = g.sum(0)
            return g.reshape(shape)

        return (d_means, fold(d_col, col_shape), fold(d_opa, opa_shape), fold(d_sca, sca_shape),
                fold(d_rot, rot_shape), None, None)


def rasterize_gaussians_batch(means3D, colors_precomp, opacities, scales, rotations, raster_settings):
    """Batched rasterization (see _RasterizeGaussiansBatch)."""
    tensors = (means3D, colors_precomp, opacities, scales, rotations)
    differentiable = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
    return _RasterizeGaussiansBatch.apply(means3D, colors_precomp, opacities, scales, rotations,
                                          raster_settings, not differentiable)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    # grad mode is always off inside Function.forward, so decide here whether anybody can
    # call backward: if not (evaluation), the overflow check is done synchronously.
    tensors = (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    differentiable = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in tensors)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, raster_settings,
                                     not differentiable)


class GaussianRasterizer(nn.Module):
    """Drop-in for diff_gaussian_rasterization.GaussianRasterizer."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        rs = self.raster_settings
        lib = _native.gsr()
        with torch.no_grad():
            P = positions.shape[0]
            pos = _f32c(positions, (P, 3))
            out = torch.empty(P, dtype=torch.uint8, device=pos.device)
            view = _f32c(rs.viewmatrix, (16,))
            proj = _f32c(rs.projmatrix, (16,))
            _native.gsr_check(lib.gsr_mark_visible(P, _ptr(pos), _ptr(view), _ptr(proj), _ptr(out),
                                                   _stream_ptr(pos.device)))
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales,
                                   rotations, cov3D_precomp, self.raster_settings)
