"""Frame-sharded data parallelism: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm; "gloo" on CPU for the logic tests).

The reference is single-process (no distributed code anywhere); its only batch
parallelism is a python loop over frames (/root/reference/model/avatar_model.py:332-365).
Frames are independent through LBS -> skin -> rasterize -> image loss, so the path shards by
frame. What couples the ranks in stage 1 is only the shared decoder output, hence:

  stage 1  every rank evaluates the (batch-invariant) net once — identical outputs and BatchNorm
           statistics everywhere; rank r renders its own frames; in backward the gradient of the
           packed per-Gaussian outputs [N,7] (residual 3, scale 1, colour 3) is exchanged with
           ONE all-reduce (5.6 MB at N=200k — latency-bound on xGMI, so a single flat message);
           the net backward + Adam step then run redundantly and identically on every rank.
           No parameter-gradient all-reduce, no SyncBN, replicas stay bit-identical.
  stage 2  decoder inputs differ per frame; parameter gradients are averaged with one flat
           all-reduce (net + pose encoder, ~19 MB). BatchNorm statistics are synchronised over the ranks
           (fused decoder: all-reduced column sums; pose encoder: SyncBatchNorm), i.e. those of the
           single-process global batch.
  pose/transl embeddings (sparse, per frame): all-gather of the sparse rows when the pose
           optimiser is active.

Losses are means over the local frames; averaging the exchanged gradients over ranks makes the
update equal to the reference's update on the global batch (equal frames per rank).

set_mode("texels") (stage 1) shards the batch-invariant decoder itself, for a FIXED global batch:
  rank r evaluates rows [r, r+1) * S^2/R of the UV map; every BatchNorm layer's column sums are all-reduced
  (11 small messages forward, 11 backward), so the statistics are those of the whole map; the packed
  per-Gaussian outputs are assembled on every rank with one all-reduce (5.6 MB) and their gradients come back
  with the one all-reduce of the frame-sharded mode; parameter gradients are partial sums over the rank's
  rows and are SUMMED once per step (net 2 MB + geometry feature map 4 MB). The same synchronised statistics
  serve stage 2 (decoder rows = the global batch's frames: SyncBN semantics of the single-process batch).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

_group = None
_enabled = False
_mode = "frames"           # stage-1 data parallelism: "frames" or "texels" (set_mode)


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun). Returns
    (rank, world_size, local_rank). A single process needs no initialisation."""
    global _enabled
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # (backend="gloo" lets the multi-rank code path be exercised on a box with fewer GPUs than ranks:
            # development only; RCCL is the production backend)
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    _enabled = world > 1
    return rank, world, local


def enable(flag: bool = True, group=None) -> None:
    global _enabled, _group
    _enabled, _group = flag, group


def set_mode(mode: str) -> None:
    """Stage-1 data parallelism: "frames" (default: every rank evaluates the batch-invariant decoder) or "texels"
    (the decoder itself is sharded by UV texels: fixed global batch), see the module docstring."""
    global _mode
    if mode not in ("frames", "texels"):
        raise ValueError(mode)
    _mode = mode


def process_group():
    """The process group of the data-parallel ranks (None = the default group)."""
    return _group


def world_size() -> int:
    return dist.get_world_size(_group) if (_enabled and dist.is_initialized()) else 1


def rank() -> int:
    return dist.get_rank(_group) if (_enabled and dist.is_initialized()) else 0


_flag_work = None
_flag_posts = []          # (work, snapshot, flag) of the reductions in flight

# Optional timing of the data-path collectives (bench.py: the per-rank breakdown of a multi-GPU line): HIP events on the
# current stream around each call, i.e. the time the stream spends in / waiting for the collective.
_timing = None


def timing_enable(on: bool = True) -> None:
    global _timing
    _timing = [] if on else None


def timing_read(reset: bool = True) -> dict:
    """{tag: (total ms, calls)} of the collectives since the last reset (blocks until their events completed)."""
    out = {}
    for s_ev, e_ev, tag in (_timing or []):
        e_ev.synchronize()
        ms, n = out.get(tag, (0.0, 0))
        out[tag] = (ms + s_ev.elapsed_time(e_ev), n + 1)
    if reset and _timing is not None:
        _timing.clear()
    return out


def _collective(tag: str, fn, tensor, **kw):
    if _timing is None or not tensor.is_cuda:
        return fn(tensor, **kw)
    s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_ev.record()
    r = fn(tensor, **kw)
    e_ev.record()
    _timing.append((s_ev, e_ev, tag))
    return r


def post_overflow_flag(device) -> None:
    """The rasterizer's backward pass raises a device-side flag on the rank whose forward pass overflowed (that
    frame's gradient is zeros, rasterizer.overflow_flag); the optimiser skips the step while it is set. Ranks must
    agree, or the replicas diverge: the flags are MAX-reduced — asynchronously, right behind the gradient
    exchange, so that the message is long done when the step needs it (wait_overflow_flag)."""
    global _flag_work
    if world_size() == 1 or not torch.device(device).type == "cuda":
        return
    from . import rasterizer
    # a SNAPSHOT is reduced, not the flag itself: a later backward pass of the same iteration (per-frame fallback path)
    # may raise the flag while this message is in flight, and the in-place result would overwrite it
    flag = rasterizer.overflow_flag(device)
    snap = flag.clone()
    _flag_work = dist.all_reduce(snap, op=dist.ReduceOp.MAX, group=_group, async_op=True)
    _flag_posts.append((_flag_work, snap, flag))


def wait_overflow_flag() -> None:
    """Order the optimiser step behind the flag's reductions (stream-level wait with RCCL, no host block) and fold their
    results into the flag."""
    global _flag_work
    for work, snap, flag in _flag_posts:
        work.wait()
        torch.maximum(flag, snap, out=flag)
    _flag_posts.clear()
    _flag_work = None


class _ExchangeGrad(torch.autograd.Function):
    """Identity in forward; in backward the incoming gradient is averaged over all ranks with a
    single all-reduce (the 'all-reduce of Gaussian-parameter grads' of BASELINE.json)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        _collective("allreduce_output_grads", dist.all_reduce, g, op=dist.ReduceOp.SUM, group=_group)
        post_overflow_flag(g.device)
        return g / dist.get_world_size(_group)


def exchange_output_grads(x: torch.Tensor) -> torch.Tensor:
    if world_size() == 1:
        return x
    return _ExchangeGrad.apply(x)


def allreduce_param_grads(params: List[torch.Tensor], average: bool = True) -> None:
    """Average (or, average=False, sum) dense parameter gradients over ranks with one flat all-reduce."""
    if world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    _collective("allreduce_param_grads", dist.all_reduce, flat, op=dist.ReduceOp.SUM, group=_group)
    if _flag_work is None:          # stage 2: no output-gradient exchange has posted the flag's reduction yet
        post_overflow_flag(flat.device)
    if average:
        flat /= dist.get_world_size(_group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allgather_sparse_grads(params: List[torch.Tensor]) -> None:
    """Sparse embedding gradients (per-frame rows) -> every rank gets every rank's rows, scaled
    by 1/world (the local losses are local means)."""
    if world_size() == 1:
        return
    W = dist.get_world_size(_group)
    for p in params:
        if p.grad is None:
            continue
        g = p.grad.coalesce() if p.grad.is_sparse else p.grad.to_sparse().coalesce()
        idx, val = g.indices(), g.values()
        n = torch.tensor([idx.shape[1]], device=val.device)
        counts = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(counts, n, group=_group)
        m = int(max(int(c) for c in counts))
        pad_i = torch.zeros(idx.shape[0], m, dtype=idx.dtype, device=idx.device)
        pad_v = torch.zeros((m,) + val.shape[1:], dtype=val.dtype, device=val.device)
        pad_i[:, :idx.shape[1]] = idx
        pad_v[:idx.shape[1]] = val
        all_i = [torch.zeros_like(pad_i) for _ in range(W)]
        all_v = [torch.zeros_like(pad_v) for _ in range(W)]
        dist.all_gather(all_i, pad_i, group=_group)
        dist.all_gather(all_v, pad_v, group=_group)
        ii = torch.cat([all_i[r][:, :int(counts[r])] for r in range(W)], dim=1)
        vv = torch.cat([all_v[r][:int(counts[r])] for r in range(W)], dim=0) / W
        p.grad = torch.sparse_coo_tensor(ii, vv, g.shape).coalesce()


def broadcast_state(modules, tensors=()) -> None:
    """Every rank takes rank 0's parameters AND buffers (BatchNorm running statistics) of `modules` and the
    extra `tensors` — one flat broadcast per dtype. Called after construction and after every checkpoint
    load, so replicas start identical whatever each process's RNG state was."""
    if world_size() == 1:
        return
    items = [t for t in tensors]
    for m in modules:
        items += [p.data for p in m.parameters()] + [b for b in m.buffers()]
    by_dtype = {}
    for t in items:
        by_dtype.setdefault(t.dtype, []).append(t)
    for group in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src=dist.get_global_rank(_group, 0) if _group is not None else 0, group=_group)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class ShardedSampler(torch.utils.data.Sampler):
    """Frames of an epoch, sharded by rank: every rank draws the SAME seeded permutation (seed + epoch
    counter, advanced on every __iter__ because the reference's loop never calls set_epoch) and keeps
    positions rank, rank+R, ... of it, truncated to a common length. On one rank it is a plain seeded
    shuffle."""

    def __init__(self, n: int, seed: int = 0, shuffle: bool = True):
        self.n, self.seed, self.shuffle, self.epoch = n, seed, shuffle, 0

    def __len__(self):
        return self.n // world_size()

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        self.epoch += 1
        order = torch.randperm(self.n, generator=g).tolist() if self.shuffle else list(range(self.n))
        R, r = world_size(), rank()
        return iter(order[r:len(self) * R:R])


def texel_sharding() -> bool:
    """Stage-1 decoder sharded by UV texels over the ranks (set_mode("texels")), see the module docstring."""
    return world_size() > 1 and _mode == "texels"


def shard_range(total: int) -> tuple:
    """Rows [r0, r1) of `total` owned by this rank (contiguous, sizes differ by at most one)."""
    W, r = world_size(), rank()
    return (total * r) // W, (total * (r + 1)) // W


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast from rank `src` of the data-parallel group."""
    if world_size() > 1:
        dist.broadcast(t, src=dist.get_global_rank(_group, src) if _group is not None else src, group=_group)
    return t


def all_reduce_sum_(t: torch.Tensor, tag: str = "allreduce_batchnorm_sums") -> torch.Tensor:
    if world_size() > 1:
        _collective(tag, dist.all_reduce, t, op=dist.ReduceOp.SUM, group=_group)
    return t


class _SumOverRanks(torch.autograd.Function):
    """value: sum of the ranks' terms (every rank sees the global number); gradient: passes to the local
    term only — the other ranks back-propagate their own terms."""

    @staticmethod
    def forward(ctx, x):
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=_group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g


def sum_over_ranks(x: torch.Tensor) -> torch.Tensor:
    return _SumOverRanks.apply(x) if world_size() > 1 else x


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def replicated_term(x: torch.Tensor) -> torch.Tensor:
    """A loss term every rank computes in full (e.g. the geometry regulariser) in a mode whose parameter
    gradients are SUMMED over ranks: the value is kept, its gradient is divided by the world size."""
    return _ScaleGrad.apply(x, 1.0 / world_size()) if world_size() > 1 else x


class _GatherSegments(torch.autograd.Function):
    """Each rank owns a contiguous slice of every segment of a flat buffer (segment j of the full buffer
    has seg[j] * n_total entries, of which this rank computed [seg[j] * n0, seg[j] * (n0 + n_local))):
    forward assembles the full buffer on every rank (zero-fill + all-reduce), backward AVERAGES the full
    gradient over ranks (local losses are means over local frames) and hands back this rank's slices."""

    @staticmethod
    def forward(ctx, local, n0, n_local, n_total, seg):
        full = local.new_zeros(sum(seg) * n_total)
        off_f, off_l = 0, 0
        for c in seg:
            full[off_f + c * n0: off_f + c * (n0 + n_local)] = local[off_l: off_l + c * n_local]
            off_f += c * n_total
            off_l += c * n_local
        _collective("allreduce_texel_records", dist.all_reduce, full, op=dist.ReduceOp.SUM, group=_group)
        ctx.dims = (n0, n_local, n_total, seg)
        return full

    @staticmethod
    def backward(ctx, g):
        n0, n_local, n_total, seg = ctx.dims
        g = g.contiguous().clone()
        _collective("allreduce_texel_record_grads", dist.all_reduce, g, op=dist.ReduceOp.SUM, group=_group)
        post_overflow_flag(g.device)
        g /= dist.get_world_size(_group)
        parts, off = [], 0
        for c in seg:
            parts.append(g[off + c * n0: off + c * (n0 + n_local)])
            off += c * n_total
        return torch.cat(parts), None, None, None, None


def gather_segments(local: torch.Tensor, n0: int, n_local: int, n_total: int, seg=(3, 1, 3)) -> torch.Tensor:
    return _GatherSegments.apply(local, n0, n_local, n_total, tuple(seg))


def barrier() -> None:
    if world_size() > 1:
        if dist.get_backend(_group) == "nccl":     # name the device: no guessing from the rank
            dist.barrier(group=_group, device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier(group=_group)


def max_over_ranks(value: float, device) -> float:
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_group)
    return float(t[0])
