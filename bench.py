#!/usr/bin/env python
"""bench.py — train iters/s (fwd+bwd) of the GaussianAvatar render-and-fit hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one stage-1 training iteration on one batch of synthetic frames per GPU:
AvatarModel.train_stage1 (LBS joint transforms -> feature net -> fused skinning -> HIP Gaussian
rasterizer) + L1 + DSSIM + regularisers (as /root/reference/train.py:70-77) -> backward -> Adam
step. Workload (BASELINE.json configs[2] / configs[3]): ~200k Gaussians, 1024x1024, batch of
2 frames per GPU (the reference's default batch size), frames sharded over GPUs with one RCCL
all-reduce of the per-Gaussian output gradients. `value` = N * K / T: reference-sized (2-frame)
iterations per second over the whole job ("weak" scaling: per-GPU work is fixed).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     the dominant rasterizer kernel: algorithmic bytes per launch / HIP-event time
  cpu_baseline the reference's PyTorch-CPU LBS + skinning + projection + L1 path (a port:
               oracle/lbs_oracle.py restates it; /root/reference is not on the GPU box),
               timed on the host cores on a bounded sample, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
EVENT_EVERY = 10          # timed steps whose dominant-family launches carry HIP events (every 20th for K >= 100)
MFMA_F32_PEAK_TF = 157.3  # dense fp32-input MFMA peak (same guide; v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (same guide; v_mfma_f32_32x32x16_bf16)
SPLIT_PRODUCTS = 6        # bf16 MFMAs per fp32 product in the split-operand decoder kernels (csrc/ganet_split.h)


def _one_pass_backward(model, frames: int) -> bool:
    """Do the hidden 128 -> 128 layers take the one-pass backward (ganet_mlp_bwd_fused)? Mirrors fused.py."""
    from gaussianavatar_amd import fused
    M = int(model.uv_coord_map.shape[0]) * frames
    return bool(fused._FUSED_BWD) and M % 32 == 0


def decoder_flops(model, frames: int = 1) -> dict:
    """Algorithmic flops (2 M N K per GEMM) of one iteration's decoder launches, by kernel family.
    Stage 1 evaluates the batch-invariant decoder once (frames=1): M = UV texels (S^2); stage 2
    evaluates it for every frame of the rank's batch: M = frames * S^2."""
    dec = model.net.decoder
    M = float(model.uv_coord_map.shape[0]) * frames
    h, cin = dec.hsize, dec.in_size
    hidden = cin * h + 3 * h * h + (h + cin) * h + 6 * h * h      # conv1, conv2-4, conv5, conv6/7 x 3 heads
    outs = h * (3 + 1 + 3)                                        # conv8 x 3 heads
    if _one_pass_backward(model, frames):
        # every hidden layer takes the one-pass kernel (data gradient + weight gradient in one launch): the ten with a
        # 128-column activated input (conv2-4, conv5's activated half, conv6/7 x 3) and the two fed by the raw input
        # (conv1, conv5's input half); the separate weight-gradient family keeps the heads (conv8 x 3)
        fused_layers = 10 * h * h + 2 * cin * h
        return {
            "mlp_fwd": 2.0 * M * (hidden + outs),
            "layer_bwd": 4.0 * M * fused_layers,
            "wgrad_act": 2.0 * M * outs,
            "head_bwd": 2.0 * M * outs,
        }
    return {
        "mlp_fwd": 2.0 * M * (hidden + outs),
        "wgrad_act": 2.0 * M * (hidden + outs),
        # input gradients: every hidden GEMM except conv1's activation operand (there is none)
        "mlp_bwd_data": 2.0 * M * hidden,
        "head_bwd": 2.0 * M * outs,
    }


def decoder_bytes(model, frames: int = 1) -> dict:
    """Algorithmic HBM bytes (fp32, every operand of every launch moved once: activations are 134 MB at
    M = 262,144 and do not survive between launches) of one iteration's decoder launches, by kernel
    family — the operand lists of gaussianavatar_amd/fused.py::_DecoderFn. Columns per row of M.
    "backward_minimum": the bytes the backward pass would move if every layer touched each of its tensors once
    (G, z, z_src in, G_src out per hidden layer; the three conv6 branches adding into conv5's gradient without
    re-reading it) — the figure the per-launch sums of the backward families are to be compared with."""
    dec = model.net.decoder
    M = float(model.uv_coord_map.shape[0]) * frames
    h, cin, xp = dec.hsize, dec.in_size, 72          # xp: decoder input padded to 8-float blocks
    outs = (3, 1, 3)
    fwd = (xp + h) + 3 * (2 * h) + (xp + 2 * h) + 6 * (2 * h) + sum(h + o for o in outs)
    head = sum(o + 2 * h for o in outs)
    # minimum: conv8 wgrad rides on head_bwd's operands (g, z7: counted in `head`); per head conv7 (G7, z7, z6 -> G6)
    # and conv6 (G6, z6 [z5 shared] ...); conv5 (G5, z5, z4, x -> G4, dx); conv4..2; conv1 (G1, z1, x -> dx)
    minimum = (head + 3 * 4 * h + (3 * 2 * h + 2 * h) + (3 * h + xp + h + cin) + 3 * 4 * h + (2 * h + xp + cin))
    if _one_pass_backward(model, frames):
        layer = (3 * 4 * h                       # conv7 -> G6: G, z, src z in, out
                 + (4 * h) + (5 * h) + (5 * h)   # conv6 -> G5: first writes, then accumulates (out read + written)
                 + 4 * h                         # conv5's activated half -> G4
                 + 3 * 4 * h                     # conv4..2
                 + (2 * h + 2 * xp)              # conv5's input half: G, z, x in, d(input) out
                 + (2 * h + 3 * xp))             # conv1: G, z, x in, d(input) read + written
        wgrad = sum(o + h for o in outs)                          # conv8 x 3
        return {"mlp_fwd": 4.0 * M * fwd, "layer_bwd": 4.0 * M * layer, "wgrad_act": 4.0 * M * wgrad,
                "head_bwd": 4.0 * M * head, "backward_minimum": 4.0 * M * minimum}
    wgrad = (sum(o + h for o in outs)            # conv8: raw g + x
             + 6 * 3 * h                         # conv7, conv6 per head: G, z, x
             + 3 * h + (2 * h + xp)              # conv5: activated operand / input operand
             + 3 * 3 * h + (2 * h + xp))         # conv4..2, conv1
    bwd = (3 * 4 * h                             # conv7 -> G6: G, z, src z, out
           + (3 * h) + (4 * h) + (5 * h)         # conv6 -> G5: first writes, then accumulates, last adds src
           + (2 * h + cin) + 4 * h               # conv5 -> d(input), G4
           + 3 * 4 * h                           # conv4..2
           + (2 * h + 2 * cin))                  # conv1 -> d(input), accumulated
    return {"mlp_fwd": 4.0 * M * fwd, "wgrad_act": 4.0 * M * wgrad, "mlp_bwd_data": 4.0 * M * bwd,
            "head_bwd": 4.0 * M * head, "backward_minimum": 4.0 * M * minimum}


def algorithmic_bytes(P: int, D: float, npix: int) -> dict:
    """Compulsory fp32 traffic per launch, each datum moved once, split by
    kernel so that a kernel's time is priced against its own bytes:
      preprocess      92 P   (read 56 = xyz12+scale12+rot16+opac4+rgb12, write 36)
      binning         16 D   (write 12/pair, read 4/pair)          [scan+scatter+sort]
      render_fwd      40 D + 20 Npix
      render_bwd      44 D + 20 Npix + 36 P   (pixels in: dL 12 + T 4 + n 4; pairs: idx 4 + 40;
                                               per-Gaussian screen-space grads written once: 36)
      preprocess_bwd  132 P  (read 56 + 36, write 40)
    Sum = raster fwd 92P+56D+20Npix, raster bwd 168P+44D+20Npix as in SURVEY.md."""
    return {
        "preprocess": 92.0 * P,
        "binning": 16.0 * D,
        "render_fwd": 40.0 * D + 20.0 * npix,
        "render_bwd": 44.0 * D + 20.0 * npix + 36.0 * P,
        "preprocess_bwd": 132.0 * P,
    }


def cpu_baseline(N: int, B: int, budget_s: float = 16.0) -> dict:
    """The reference's CPU path restated in oracle/lbs_oracle.py, same N and B as the GPU workload, on
    the host cores. Timed in TWO forms: `as_the_reference_runs_it` — the whole of lbs() on an
    SMPL-sized body (6,890 vertices: shape and pose blend shapes, per-vertex blend of the joint transforms, vertex
    skinning, /root/reference/submodules/smplx/lbs.py:206-247) whose vertices the path then discards — and
    `dead_work_removed` (rest joints precomputed, joint transforms only). Each: the median of 5 windows with their
    spread; `value` is the first form."""
    import statistics
    from oracle import lbs_oracle as O
    from gaussianavatar_amd.synthetic import make_assets, make_frames
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    assets = make_assets(num_points=min(N, 64 * 64 * 3 // 4), uv_size=64)        # skeleton, poses, camera
    frames = make_frames(assets, B, 1024, 1024)
    J = assets["joints_rest"]
    parents = torch.tensor(assets["parents"], dtype=torch.long)
    inv = torch.linalg.inv(assets["cano_joint_mat"]).expand(B, -1, -1, -1)
    pts = (torch.randn(1, N, 3, generator=g) * 0.4).expand(B, -1, -1)
    w = torch.rand(N, 24, generator=g) ** 8
    w = (w / w.sum(1, keepdim=True))[None].expand(B, -1, -1).contiguous()
    full_proj = torch.tensor(frames["camera"]["full_proj_transform"])
    pose0, transl = frames["pose"][:B].clone(), frames["transl"][:B].clone()
    # an SMPL-sized body model (seeded; the shapes of /root/reference/submodules/smplx/body_models.py:127-138)
    V, Jn = 6890, parents.shape[0]
    jr = torch.rand(Jn, V, generator=g) ** 16
    lw = torch.rand(V, Jn, generator=g) ** 8
    body = dict(v_template=torch.randn(V, 3, generator=g) * torch.tensor([0.3, 0.8, 0.15]),
                shapedirs=torch.randn(V, 3, 10, generator=g) * 0.01,
                posedirs=torch.randn((Jn - 1) * 9, V * 3, generator=g) * 0.001,
                J_regressor=jr / jr.sum(1, keepdim=True), lbs_weights=lw / lw.sum(1, keepdim=True))
    betas = torch.zeros(B, 10)

    def one(full: bool):
        pose = pose0.clone().requires_grad_(True)
        res = (torch.zeros(B, N, 3)).requires_grad_(True)
        O.cpu_baseline_step(pose, transl, J, parents, inv, pts, res, w, full_proj, body=body if full else None,
                            betas=betas if full else None)

    # torch's intra-op threading degrades badly when every hardware thread of a large host is
    # used for these small ops (256 threads: 9 s per iteration); take the best of a few counts
    best = None
    tried = {}
    for nt in sorted({min(cores, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        one(True)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            one(True)
            ts.append(time.perf_counter() - t0)
        dt = statistics.median(ts)
        tried[nt] = round(1.0 / dt, 1)
        if best is None or dt < best[1]:
            best = (nt, dt)
    threads = best[0]
    torch.set_num_threads(threads)

    def windows(full: bool, n_win=5):
        one(full)
        rates, total, t_all = [], 0, time.perf_counter()
        for _ in range(n_win):
            t0, n = time.perf_counter(), 0
            while True:
                one(full)
                n += 1
                if time.perf_counter() - t0 >= budget_s / (2 * n_win) or n >= 80:
                    break
            rates.append(n / (time.perf_counter() - t0))
            total += n
        return {"iters_per_s_median": statistics.median(rates), "iters_per_s_min": min(rates), "iters_per_s_max": max(rates),
                "windows": n_win, "iterations": total, "wall_s": time.perf_counter() - t_all}

    as_ref, lean = windows(True), windows(False)
    return {"value": as_ref["iters_per_s_median"], "unit": "iters/s", "cores": threads, "kind": "port",
            "as_the_reference_runs_it": as_ref, "dead_work_removed": lean,
            "sample": f"median of {as_ref['windows']} windows ({as_ref['iterations']} fwd+bwd iterations, {as_ref['wall_s']:.1f} s) of the "
                      f"reference's CPU path as it runs it: lbs() on a 6,890-vertex body (blend shapes + vertex skinning, "
                      f"discarded) -> joint transforms -> skinning of N={N} points -> projection -> L1-to-black, B={B} frames "
                      f"(no rasterizer/net exists on the CPU side); spread {as_ref['iters_per_s_min']:.1f}-{as_ref['iters_per_s_max']:.1f}; "
                      f"with the dead vertex work removed: {lean['iters_per_s_median']:.1f} "
                      f"({lean['iters_per_s_min']:.1f}-{lean['iters_per_s_max']:.1f}); {threads} of {cores} host threads; "
                      f"threads tried -> it/s: {tried}; CPU: {_cpu_model()}",
            "threads_tried_iters_per_s": tried}


def plumbing_config1(budget_s: float = 10.0) -> dict:
    """BASELINE.json configs[0]: 5k random Gaussians, 256x256, SMPL T-pose, PyTorch-CPU LBS + projection only
    (no rasterizer), L1 to black, fwd+bwd — the reference's CPU path (restated in
    oracle/lbs_oracle.py), timed on the host cores. Needs no HIP device."""
    from oracle import lbs_oracle as O
    from gaussianavatar_amd.camera import test_pose_camera
    from gaussianavatar_amd.synthetic import make_assets
    N, B = 5000, 1
    cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    assets = make_assets(num_points=3000, uv_size=64)
    J = assets["joints_rest"]
    parents = torch.tensor(assets["parents"], dtype=torch.long)
    inv = torch.linalg.inv(assets["cano_joint_mat"]).expand(B, -1, -1, -1)
    pts = (torch.randn(1, N, 3, generator=g) * 0.4).expand(B, -1, -1)
    w = torch.rand(N, 24, generator=g) ** 8
    w = (w / w.sum(1, keepdim=True))[None].expand(B, -1, -1).contiguous()
    full_proj = torch.tensor(test_pose_camera(256)["full_proj_transform"])
    transl = torch.zeros(B, 3)

    def one():
        pose = torch.zeros(B, 72, requires_grad=True)            # T-pose
        res = torch.zeros(B, N, 3, requires_grad=True)
        return O.cpu_baseline_step(pose, transl, J, parents, inv, pts, res, w, full_proj)

    best = None
    for nt in sorted({min(cores, c) for c in (1, 4, 8, 16)}):
        torch.set_num_threads(nt)
        one()
        t0 = time.perf_counter()
        for _ in range(5):
            one()
        dt = (time.perf_counter() - t0) / 5
        if best is None or dt < best[1]:
            best = (nt, dt)
    torch.set_num_threads(best[0])
    t0 = time.perf_counter()
    n = 0
    while True:
        loss = one()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 5000:
            break
    return {"metric": "config-1 plumbing: PyTorch-CPU LBS + projection fwd+bwd iters/s (no rasterizer)",
            "value": n / el, "unit": "iters/s", "n_gpus": 0, "steps": n, "warmup": 6, "ms_per_step": 1e3 * el / n,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[0]: 5k Gaussians, 256x256, SMPL T-pose, CPU LBS + skinning + "
                                   "projection + L1-to-black only", "gaussians": N, "image": [256, 256],
                       "final_loss": float(loss.detach())},
            "cpu_baseline": {"value": n / el, "unit": "iters/s", "cores": best[0], "kind": "port",
                             "sample": f"{n} iterations, {el:.1f} s wall, {best[0]} of {cores} host threads; CPU: {_cpu_model()}"}}


_TRAFFIC_FAMILIES = {   # kernel-name substring -> family key (tools/traffic_json.py)
    "layer_bwd_spec_kernel<false, true, 128>": "layer_bwd", "render_bwd_kernel": "render_bwd",
    "layer_fwd_spec_kernel<1, false>": "layer_fwd", "render_fwd_kernel": "render_fwd",
}


def _counter_means(csv_path, counter):
    """{family: mean of `counter` over the second half of the family's launches} from a rocprofv3 counter_collection CSV."""
    import csv
    vals = {}
    with open(csv_path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            for sub, fam in _TRAFFIC_FAMILIES.items():
                if sub in r["Kernel_Name"]:
                    vals.setdefault(fam, []).append(float(r["Counter_Value"]))
                    break
    return {fam: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for fam, v in vals.items()}


def measure_traffic(args):
    """HBM-side traffic per launch of the families the bench line prices, measured NOW: two short rocprofv3 passes of this
    same workload (--pmc FETCH_SIZE, --pmc WRITE_SIZE: the TCC counters do not fit one pass; with --kernel-trace only),
    mean over the second half of each kernel's launches. Returns {family: {fetch_kb, write_kb}} or None (rocprofv3 missing,
    a pass failed or timed out: the caller then cites the committed summary)."""
    import glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    # this process is itself being profiled (rocprofv3 -- python bench.py ...): no profiler inside a profiler
    if any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "6", "--warmup", "4",
             "--points", str(args.points), "--size", str(args.size), "--height", str(args.height),
             "--frames-per-gpu", str(args.frames_per_gpu), "--stage", str(args.stage), "--smpl-type", args.smpl_type,
             "--uv", str(args.uv), "--iteration", str(args.iteration), "--no-cpu-baseline", "--no-kernel-events",
             "--no-fixed-batch", "--no-secondary", "--no-measure-traffic"]
    out = {}
    tmp = tempfile.mkdtemp(prefix="ga_traffic_", dir="/tmp")
    try:
        for counter, key in (("FETCH_SIZE", "fetch_kb"), ("WRITE_SIZE", "write_kb")):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + child
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            try:
                proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                        start_new_session=True)
            except OSError:
                return None
            try:
                rc = proc.wait(timeout=120)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)      # the profiler AND the bench it runs (its own session / group)
                except OSError:
                    pass
                proc.wait()
                return None
            if rc != 0:
                return None
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None
            for fam, mean in _counter_means(files[0], counter).items():
                out.setdefault(fam, {})[key] = mean
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: v for k, v in out.items() if "fetch_kb" in v and "write_kb" in v} or None


def _traffic_file():
    """The newest committed PMC-traffic summary (profiles/rNN_traffic.json; tools/pmc_traffic.sh writes them)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    return files[-1] if files else None


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def fixed_global_batch_line(args, dev, rank, world, global_batch=8):
    """The same stage-1 workload with a FIXED global batch of 8 frames (BASELINE.json configs[3]) spread over the
    ranks — decoder sharded by UV texels, synchronised BatchNorm statistics — as a second, shorter measurement, so
    that one driver run carries both the weak value (`value`) and a strong-scaling value."""
    from gaussianavatar_amd import parallel, rasterizer
    from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
    from gaussianavatar_amd.losses import l1_loss_w, ssim, weighted_sum
    if global_batch % world:
        return None
    B = global_batch // world
    prev_mode = "texels" if parallel.texel_sharding() else "frames"
    parallel.set_mode("texels")
    torch.manual_seed(0)
    mp, npar, op = default_params(batch_size=B, num_points=args.points, image_width=args.size,
                                  image_height=args.height or args.size, num_frames=max(16, global_batch),
                                  train_stage=1, smpl_type=args.smpl_type,
                                  query_posmap_size=args.uv or (1024 if args.points > 512 * 512 else 512))
    model = AvatarModel(mp, npar, op, train=True, device=dev)
    model.training_setup()
    model.net.train()
    ds = model.train_dataset
    W, H = args.size, args.height or args.size
    gt = torch.ones(B, 3, H, W, device=dev)
    gt[:, :, H // 5: 4 * H // 5, 2 * W // 5: 3 * W // 5] = 0.6
    batches = [collate_frames([ds[(s * global_batch + rank * B + k) % len(ds)] for k in range(B)], dev) for s in range(2)]
    l = op.lambda_dssim

    def step(i):
        image, points, offset_loss, geo_loss, scale_loss = model.train_stage1(batches[i % 2], args.iteration)
        loss = weighted_sum([scale_loss, offset_loss, l1_loss_w(image, gt), ssim(image, gt), geo_loss],
                            [op.lambda_scale, op.lambda_rgl, 1.0 - l, -l, 1.0], bias=l)
        model.zero_grad(1)
        loss.backward()
        model.step(1)

    # (secondary measurement: its own warm-up, whatever --warmup says — on a fresh box the first iterations of a new
    # model in the same process still grow the allocator's pools and the rasterizer's pair capacity: with 5 warm-up steps the
    # 20 timed ones read 107-145 it/s, with 20 they read the steady 200)
    steps, warm = max(20, min(args.steps, 50)), max(20, min(args.warmup, 30))
    for i in range(warm):
        step(i)
    rasterizer.check_overflow(block=True)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    parallel.barrier()
    torch.cuda.synchronize()
    el = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    parallel.set_mode(prev_mode)
    return {"global_batch": global_batch, "frames_per_gpu": B, "steps": steps, "warmup": warm,
            "value": steps / el, "unit": "iters/s (one iteration = the 8-frame global batch)",
            "frames_per_s": global_batch * steps / el, "ms_per_step": 1e3 * el / steps, "scaling": "strong",
            "parallelism": (f"frames sharded over dp{world}; decoder sharded by UV texels, synchronised BatchNorm "
                            f"statistics, outputs assembled with one all-reduce" if world > 1 else "single GPU")}


def secondary_stage2_line(args, dev):
    """A second, shorter measurement every default bench line carries (`secondary`): the stage-2 iteration
    (pose-encoder UNet on, the decoder evaluated per frame: M = frames x S^2 rows) at the headline size — the workload
    of BASELINE.json configs[4] on SMPL, one GPU. Same loop as `--stage 2` (/root/reference/train.py:78-86)."""
    from gaussianavatar_amd import rasterizer
    from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
    from gaussianavatar_amd.losses import l1_loss_w, ssim, weighted_sum
    B = args.frames_per_gpu
    torch.manual_seed(0)
    mp, npar, op = default_params(batch_size=B, num_points=args.points, image_width=args.size,
                                  image_height=args.height or args.size, num_frames=16, train_stage=2,
                                  smpl_type=args.smpl_type,
                                  query_posmap_size=args.uv or (1024 if args.points > 512 * 512 else 512))
    model = AvatarModel(mp, npar, op, train=True, device=dev)
    model.training_setup()
    model.net.train()
    with torch.no_grad():      # stand-in for the stage-1 checkpoint stage 2 starts from (see main())
        model.net.decoder.conv8N.weight.mul_(0.01)
        model.net.decoder.conv8N.bias.fill_(-5.65)
    ds = model.train_dataset
    W, H = args.size, args.height or args.size
    gt = torch.ones(B, 3, H, W, device=dev)
    gt[:, :, H // 5: 4 * H // 5, 2 * W // 5: 3 * W // 5] = 0.6
    batches = [collate_frames([ds[(s * B + k) % len(ds)] for k in range(B)], dev) for s in range(2)]
    l = op.lambda_dssim

    def step(i):
        image, points, pose_loss, offset_loss = model.train_stage2(batches[i % 2], args.iteration)
        loss = weighted_sum([offset_loss, l1_loss_w(image, gt), ssim(image, gt), pose_loss],
                            [op.lambda_rgl, 1.0 - l, -l, 10.0], bias=l)
        model.zero_grad(1)
        loss.backward()
        model.step(1)

    # own warm-up (see fixed_global_batch_line); stage 2 needs more of it: in the first GPU process on a fresh box the
    # first ~50 iterations of this model are host-bound (11 ms instead of 7.7 ms per iteration; a 300-step run reads
    # 131-134 it/s in every 50-step window after the first 50 iterations)
    steps, warm = max(20, min(args.steps, 30)), 60
    for i in range(warm):
        step(i)
    rasterizer.check_overflow(block=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    N = int(model.query_points.shape[1])
    return {"workload": f"stage-2 train iteration (pose-encoder UNet + per-frame decoder, M = {B} x {int(model.uv_coord_map.shape[0])} "
                        f"rows), {N} Gaussians, {W}x{H}, {B} frames, {args.smpl_type.upper()}-shaped body, 1 GPU "
                        f"(the single-GPU share of BASELINE.json configs[4]'s stage 2; `bench.py --stage 2`)",
            "value": steps / el, "unit": "iters/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warm}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--points", type=int, default=200_000)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--frames-per-gpu", type=int, default=2)
    ap.add_argument("--height", type=int, default=0, help="image height if it differs from --size (e.g. 1080)")
    ap.add_argument("--stage", type=int, default=1, choices=(1, 2),
                    help="2 = stage-2 iteration (pose-encoder UNet on, per-frame decoder input); secondary "
                         "workload of BASELINE.json configs[4], the headline metric is stage 1")
    ap.add_argument("--smpl-type", default="smpl", choices=("smpl", "smplx"))
    ap.add_argument("--uv", type=int, default=0, help="query UV map edge (default: 512, or 1024 when --points > 262144)")
    ap.add_argument("--iteration", type=int, default=7,
                    help="training iteration passed to train_stage1: the scale warm-up (1e-3 x iteration, "
                         "/root/reference/model/avatar_model.py:316) sets the Gaussians' size at random init — "
                         "5 / 7 / 14 give ~2.5 / 3.5 / 7 mm, i.e. ~2.8 P / 4 P / 9 P tile pairs (D sensitivity)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fix the GLOBAL batch (frames per iteration over all GPUs) instead of frames per GPU: "
                         "strong scaling, BASELINE.json configs[3] = 8")
    ap.add_argument("--config", type=int, default=0, choices=(0, 1, 2, 3, 4, 5),
                    help="BASELINE.json configs[N-1]: 1 CPU plumbing (no HIP device needed), 2 100k@512^2, 3 headline, "
                         "4 headline with a fixed global batch of 8 frames, 5 stage 2 / SMPL-X / 300k / 1920x1080 / global batch 8")
    ap.add_argument("--dp-mode", default="", choices=("", "frames", "texels"),
                    help="stage-1 data parallelism (gaussianavatar_amd/parallel.py): frames = every rank evaluates the "
                         "batch-invariant decoder (default for weak scaling); texels = the decoder is sharded by UV "
                         "texels (default with --global-batch: fixed total work)")
    ap.add_argument("--series", type=int, default=0,
                    help="also report iters/s per window of this many timed steps (HIP events on the loop's stream: "
                         "sustained vs burst rate) under config.series")
    ap.add_argument("--dist-backend", default="", choices=("", "nccl", "gloo"),
                    help="development: gloo exercises the multi-rank path on a box with fewer GPUs than ranks")
    ap.add_argument("--share-device0", action="store_true", help="development: every rank on GPU 0")
    ap.add_argument("--no-fixed-batch", action="store_true",
                    help="skip the second, shorter measurement with a fixed global batch of 8 frames (fixed_global_batch)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second short measurement of the default line: the stage-2 iteration (secondary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="do not re-measure roofline.traffic with two short rocprofv3 --pmc passes of this workload after "
                         "the timed run (N = 1, when rocprofv3 is on PATH; ~1 min); the committed profiles/rNN_traffic.json is cited instead")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket rasterizer kernels with HIP events in the timed region")
    args = ap.parse_args()
    if args.config == 1:
        print(json.dumps(plumbing_config1()))
        return
    if args.config == 2:
        args.points, args.size = 100_000, 512
    elif args.config == 4:
        args.global_batch = args.global_batch or 8
    elif args.config == 5:
        args.stage, args.smpl_type, args.points, args.size, args.height = 2, "smplx", 300_000, 1920, 1080
        args.global_batch = args.global_batch or 8

    from gaussianavatar_amd import parallel
    if args.stage == 1 and (args.dp_mode or args.global_batch):
        parallel.set_mode(args.dp_mode or "texels")
    rank, world, local = parallel.init_from_env(backend=args.dist_backend or None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if args.share_device0:       # development: several ranks on one GPU (with --dist-backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from gaussianavatar_amd import _native, fused, rasterizer
    from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
    from gaussianavatar_amd.losses import l1_loss_w, ssim, weighted_sum

    torch.manual_seed(0)      # (replicas are synchronised from rank 0 anyway: AvatarModel.sync_replicas)
    if args.global_batch:
        assert args.global_batch % world == 0, "--global-batch must be a multiple of the GPU count"
        args.frames_per_gpu = args.global_batch // world
    B = args.frames_per_gpu
    uv = args.uv or (1024 if args.points > 512 * 512 else 512)
    mp, npar, op = default_params(batch_size=B, num_points=args.points, image_width=args.size,
                                  image_height=args.height or args.size, num_frames=max(16, B * world),
                                  train_stage=args.stage, smpl_type=args.smpl_type, query_posmap_size=uv)
    model = AvatarModel(mp, npar, op, train=True, device=dev)
    model.training_setup()
    model.net.train()
    ds = model.train_dataset
    W = args.size
    H = args.height or args.size
    # target images: white background (as the reference composites) with a grey silhouette band
    gt = torch.ones(B, 3, H, W, device=dev)
    gt[:, :, H // 5: 4 * H // 5, 2 * W // 5: 3 * W // 5] = 0.6
    nf = len(ds)
    batches = []
    for s in range(4):        # a few distinct batches; rank r takes frames r*B.. of each global batch
        ids = [(s * world * B + rank * B + k) % nf for k in range(B)]
        batches.append(collate_frames([ds[i] for i in ids], dev))
    epoch, iteration = 1, args.iteration   # iteration 7 < 1000: scale warm-up gives ~3.5 mm Gaussians at init
    if args.stage == 2:
        # the reference's stage 2 starts from a trained stage-1 checkpoint (no scale warm-up,
        # /root/reference/model/avatar_model.py:416); a random-init scale head would give 0.5 m
        # Gaussians (every Gaussian on every tile). Stand-in: scale head biased to ~3.5 mm.
        with torch.no_grad():
            model.net.decoder.conv8N.weight.mul_(0.01)
            model.net.decoder.conv8N.bias.fill_(-5.65)

    def step(i):
        batch = batches[i % len(batches)]
        if args.stage == 1:            # /root/reference/train.py:68-76
            image, points, offset_loss, geo_loss, scale_loss = model.train_stage1(batch, iteration)
            # loss = lambda_scale*scale + lambda_rgl*offset + (1-l)*L1 + l*(1 - SSIM) + geo, composed in one
            # launch (losses.weighted_sum) instead of one zero-dimensional kernel per operator
            l = op.lambda_dssim
            loss = weighted_sum([scale_loss, offset_loss, l1_loss_w(image, gt), ssim(image, gt), geo_loss],
                                [op.lambda_scale, op.lambda_rgl, 1.0 - l, -l, 1.0], bias=l)
        else:                          # /root/reference/train.py:78-86
            image, points, pose_loss, offset_loss = model.train_stage2(batch, iteration)
            l = op.lambda_dssim
            loss = weighted_sum([offset_loss, l1_loss_w(image, gt), ssim(image, gt), pose_loss],
                                [op.lambda_rgl, 1.0 - l, -l, 10.0], bias=l)
        model.zero_grad(epoch)
        loss.backward()
        model.step(epoch)
        return loss

    # Kernel timing. Bracketing EVERY launch with HIP events costs ~1.4 ms per iteration (~100
    # launches x 2 event packets), so: the last few WARM-UP steps run fully instrumented (complete
    # per-kernel table + which kernel family dominates the iteration); in the TIMED region only that
    # dominant family is bracketed, and only on every EVENT_EVERY-th step (an event pair costs ~40 us of
    # lost launch overlap, 12 launches/iteration would still be ~6 %) — `roofline` is computed from those.
    events = not args.no_kernel_events
    probe = min(3, args.warmup) if events else 0
    for i in range(args.warmup - probe):
        step(i)
    probe_r, probe_n = {}, {}
    probe_pairs = 0.0
    if probe:
        rasterizer.check_overflow(block=True)
        rasterizer.pair_statistics(reset=True)
        rasterizer.profile_enable(True)
        fused.profile_enable(True)
        rasterizer.profile_read(reset=True)
        fused.profile_read(reset=True)
        parallel.timing_enable(True)
        pev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        pev[0].record()
        for i in range(args.warmup - probe, args.warmup):
            step(i)
        pev[1].record()
        probe_r = rasterizer.profile_read(reset=True)
        probe_n = fused.profile_read(reset=True)
        probe_c = parallel.timing_read()
        parallel.timing_enable(False)
        pev[1].synchronize()
        probe_ms = pev[0].elapsed_time(pev[1]) / probe
        # the scene of the instrumented steps (with --iteration far from 7 the model is still shrinking its Gaussians
        # during the warm-up: the per-kernel table belongs to THIS pair count, not to the timed region's)
        _pn, probe_pairs = rasterizer.pair_statistics(reset=True)
        rasterizer.profile_enable(False)
        fused.profile_enable(False)
    rasterizer.check_overflow(block=True)
    rasterizer.pair_statistics(reset=True)
    share = {k: ms / probe for k, (ms, n) in list(probe_r.items()) + list(probe_n.items()) if n} if probe else {}
    dom_family = max(share, key=share.get) if share else None
    dom_lib = rasterizer if dom_family in probe_r else (fused if dom_family in probe_n else None)
    sampled_steps = 0
    every = EVENT_EVERY * (2 if args.steps >= 100 else 1)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(args.steps):
        if args.series and i % args.series == 0:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
        sample = dom_lib is not None and i % every == 0
        if sample:
            dom_lib.profile_enable([dom_family])
            sampled_steps += 1
        loss = step(args.warmup + i)
        if sample:
            dom_lib.profile_enable(False)
    parallel.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, dev)
    series = None
    if args.series and len(marks) > 1:
        series = [round(args.series / (a.elapsed_time(b) * 1e-3), 2) for a, b in zip(marks, marks[1:])]
    timed = dom_lib.profile_read(reset=True) if dom_lib is not None else {}
    rasterizer.profile_enable(False)
    fused.profile_enable(False)
    ncalls, mean_pairs = rasterizer.pair_statistics(reset=True)
    final_loss = float(loss.detach())
    backend = torch.distributed.get_backend() if (world > 1 and torch.distributed.is_initialized()) else None
    if world > 1 and not args.dist_backend:
        assert backend == "nccl", f"multi-GPU bench must run over RCCL (backend nccl), got {backend}"
    dflops = decoder_flops(model, 1 if args.stage == 1 else B)
    dbytes = decoder_bytes(model, 1 if args.stage == 1 else B)
    fixed, secondary = None, None
    headline = args.stage == 1 and not args.global_batch and args.config in (0, 3)
    Nq = model.query_points.shape[1]
    if headline and not (args.no_fixed_batch and (args.no_secondary or world > 1)):
        del model, batches
        torch.cuda.empty_cache()
        if not args.no_fixed_batch:
            fixed = fixed_global_batch_line(args, dev, rank, world)
            torch.cuda.empty_cache()
        if world == 1 and not args.no_secondary:
            secondary = [secondary_stage2_line(args, dev)]

    if rank != 0:
        return
    N = Nq
    # weak scaling (default): every GPU renders `frames_per_gpu` frames, value counts reference-sized
    # iterations (one per GPU and step). --global-batch: the job does ONE iteration of that many frames per
    # step whatever the GPU count ("strong"): value = steps / time.
    value = (1 if args.global_batch else world) * args.steps / elapsed
    out = {
        "metric": "train iters/s (fwd+bwd), 200k Gaussians @1024^2, 1/2/4/8 MI355X",
        "value": value, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if args.global_batch else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "ranks": world, "dist_backend": backend, **({"fixed_global_batch": fixed} if fixed else {}),
        **({"secondary": secondary} if secondary else {}),
        "config": {"workload": f"stage-{args.stage} train iteration (LBS + feature net + skinning + Gaussian rasterizer "
                               f"fwd+bwd + L1/DSSIM + Adam), {N} Gaussians, {W}x{H}, {B} frames per GPU "
                               + ("(BASELINE.json configs[2])" if (args.stage, N, W, H) == (1, 200_000, 1024, 1024)
                                  else "(secondary workload)")
                               + f"; synthetic {args.smpl_type.upper()}-shaped body, random-init net",
                   "gaussians": N, "image": [H, W], "uv_map": uv, "iteration": iteration, "frames_per_gpu": B,
                   "global_batch": B * world, "frames_per_s": B * world * args.steps / elapsed,
                   "parallelism": (f"frames sharded over dp{world}; decoder sharded by UV texels (synchronised BatchNorm "
                                   f"statistics, outputs assembled with one all-reduce, parameter gradients summed)"
                                   if parallel.texel_sharding() else
                                   f"frame-sharded dp{world}, one all-reduce of [N,7] output grads"
                                   + ("" if args.stage == 1 else ", synchronised BatchNorm, parameter gradients averaged")),
                   "mean_tile_pairs_per_frame": mean_pairs, "final_loss": final_loss,
                   **({"series": {"window_steps": args.series, "iters_per_s": series}} if series else {}),
                   "decoder_gemm_arithmetic": (
                       "fp32 in / fp32 out; operands split exactly into three bf16 pieces, six bf16 MFMA products per "
                       "fp32 product accumulated in fp32 (csrc/ganet_split.h; error vs float64 ~4e-7 of the tensor's "
                       "max, tests/test_fused_gpu.py::test_split_mfma_is_fp32_accurate)")},
    }
    # HBM-side traffic (PMC counters) of the priced families: re-measured right now when rocprofv3 is here, else the
    # committed summary of the same command is cited (labelled either way)
    fresh_traffic = None
    if probe and world == 1 and rank == 0 and not args.no_measure_traffic:
        torch.cuda.synchronize()
        fresh_traffic = measure_traffic(args)
    fresh_note = ("two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of this workload, run by this "
                  "invocation right after its timed region (bench.measure_traffic: 6 steps each, mean over the second half of "
                  "the launches); 2*FETCH+WRITE (gfx950 wide-read correction)")
    if probe:
        # one launch of every rasterizer kernel processes all B frames of the rank's batch
        # (priced at the pair count of the steps the events come from)
        alg = {k: v * B for k, v in algorithmic_bytes(N, probe_pairs or mean_pairs, H * W).items()}
        stage_of = {"preprocess": "preprocess", "tile_scan": "binning", "scatter": "binning", "tile_sort": "binning",
                    "render_fwd": "render_fwd", "render_bwd": "render_bwd", "preprocess_bwd": "preprocess_bwd"}

        # the decoder GEMMs' matrix roof: exact fp32 on the bf16 pipe costs 6 bf16 MFMA flops per algorithmic flop,
        # so the algorithmic-flop peak is 2500 / 6 TFLOP/s
        split = True
        mfma_peak = MFMA_BF16_PEAK_TF / SPLIT_PRODUCTS

        def describe(name, ms, n, iters):
            """per-kernel-family record: time, and its algorithmic work priced against its roofline"""
            d = {"launches_per_iter": n / iters, "avg_us": 1e3 * ms / n, "us_per_iter": 1e3 * ms / iters}
            if name in dflops and name in dbytes:
                # both roofs: the one that demands more time is the bound
                tf = dflops[name] / (d["us_per_iter"] * 1e-6) / 1e12
                gbs = dbytes[name] / (d["us_per_iter"] * 1e-6) / 1e9
                mf, hf = tf / mfma_peak, gbs / HBM_PEAK_GBS
                d.update({"bound": "mfma" if mf >= hf else "hbm", "flops_per_iter": dflops[name], "TFLOPs": tf,
                          "frac_of_mfma_peak": mf, "bytes_per_iter": dbytes[name], "GBps": gbs,
                          "frac_of_hbm_peak": hf, "frac_of_peak": max(mf, hf)})
            elif name in ("render_fwd", "render_bwd", "preprocess", "preprocess_bwd"):
                gbs = alg[name] / (d["us_per_iter"] * 1e-6) / 1e9
                d.update({"bound": "hbm", "algorithmic_bytes_per_launch": alg[name], "GBps": gbs,
                          "frac_of_peak": gbs / HBM_PEAK_GBS})
            return d

        kern = {k: describe(k, ms, n, probe) for k, (ms, n) in list(probe_r.items()) + list(probe_n.items()) if n}
        table = {}
        for k, st in stage_of.items():
            if k in kern:
                table.setdefault(st, {"us_per_iter": 0.0})["us_per_iter"] += kern[k]["us_per_iter"]
        for st, t in table.items():
            t["algorithmic_bytes"] = alg[st]
            t["GBps"] = alg[st] / (t["us_per_iter"] * 1e-6) / 1e9
            t["frac_of_8TBps"] = t["GBps"] / HBM_PEAK_GBS
        # ---- roofline of the dominant kernel family, from the events of the TIMED steps
        ms, n = timed[dom_family]
        d = describe(dom_family, ms, n, sampled_steps)
        if dom_family in dflops:
            # HBM bytes of one 128->128 launch of this family (PMC passes committed under profiles/)
            traffic, traffic_note = None, None
            tpath = _traffic_file()
            if fresh_traffic and dom_family in fresh_traffic:
                t = fresh_traffic[dom_family]
                traffic = (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0
                traffic_note = ("HBM bytes of one 128->128 launch (M = 262,144) of this family, measured in this run: " + fresh_note +
                                "; the algorithmic traffic of that launch is 4 x 134 MB of activations + 17 MB of partial "
                                "weight-gradient tiles")
            elif tpath:
                t = json.load(open(tpath)).get(dom_family)
                if t:
                    traffic = (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0
                    traffic_note = ("HBM bytes of one 128->128 launch (M = 262,144) of this family: rocprofv3 --pmc "
                                    f"FETCH_SIZE / WRITE_SIZE, separate passes of this command ({os.path.relpath(tpath, ROOT)}; "
                                    "tools/pmc_traffic.sh re-measures it: PMC passes cannot ride inside the timed run); "
                                    "2*FETCH+WRITE (gfx950 wide-read correction); equals the algorithmic traffic of that "
                                    "launch (4 x 134 MB of activations + 17 MB of partial weight-gradient tiles)")
            mfma = {"bound": "mfma", "achieved": d["TFLOPs"], "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": d["frac_of_mfma_peak"], "algorithmic_flops_per_iter": d["flops_per_iter"]}
            hbm = {"bound": "hbm", "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": d["frac_of_hbm_peak"], "algorithmic_bytes_per_iter": d["bytes_per_iter"]}
            # how far the backward pass is from its byte minimum (every tensor of every layer touched once)
            bw_launched = (dbytes.get("layer_bwd", dbytes.get("mlp_bwd_data", 0.0) + dbytes.get("wgrad_act", 0.0))
                           + dbytes["head_bwd"])
            byte_minimum = {"backward_minimum_bytes_per_iter": dbytes["backward_minimum"],
                            "backward_launched_bytes_per_iter": bw_launched,
                            "backward_launched_over_minimum": bw_launched / dbytes["backward_minimum"],
                            "note": "algorithmic bytes of the decoder backward's launches (layer_bwd + head_bwd) "
                                    "against the bytes it would move if every layer touched each of its tensors once; the "
                                    "excess is the three conv6 branches accumulating into conv5's gradient through HBM and "
                                    "re-reading z5 (DESIGN.md section 4.3)"}
            first, second = (mfma, hbm) if d["bound"] == "mfma" else (hbm, mfma)
            roof = {"kernel": dom_family, **first, "traffic": traffic, "traffic_note": traffic_note,
                    "avg_us": d["avg_us"], "launches_per_iter": d["launches_per_iter"], "us_per_iter": d["us_per_iter"],
                    "other_roof": second, "byte_minimum": byte_minimum,
                    "note": "a tall-skinny fp32 GEMM family (activations 134 MB per operand, 8.6 GFLOP per 128x128 "
                            "layer). Both roofs are priced and the one that demands more time is reported as the bound: "
                            "HBM (8 TB/s, algorithmic bytes) and the matrix pipe — "
                            + ("fp32 operands split exactly into three bf16 pieces, 6 v_mfma_f32_32x32x16_bf16 products "
                               "per fp32 product accumulated in fp32: algorithmic-flop peak 2500 / 6 = 416.7 TFLOP/s"
                               if split else "v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense")
                            + "; the family's launches differ in shape, so flops, bytes and time are summed over one "
                              "iteration"}
        else:
            st = stage_of.get(dom_family, dom_family)
            gbs = alg[st] / (d["us_per_iter"] * 1e-6) / 1e9
            roof = {"kernel": dom_family, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_us": d["avg_us"], "frames_per_launch": B,
                    "algorithmic_bytes_per_launch": alg[st]}
        roof["share_of_iteration"] = d["us_per_iter"] / (1e6 * elapsed / args.steps)
        roof["measured"] = (f"HIP events on the launch stream around this family's launches in {sampled_steps} of the "
                            f"{args.steps} timed steps ({n} launches)")
        out["roofline"] = roof
        # ---- the rasterizer backward (north_star's named kernel), from the instrumented warm-up steps.
        # HBM traffic per launch: PMC counters need their own rocprofv3 passes, so the value is the one
        # measured for this same command and committed under profiles/ (null if the workload differs)
        if "render_bwd" in kern:
            rb = kern["render_bwd"]
            traffic, traffic_note = None, None
            tpath = _traffic_file()
            if fresh_traffic and "render_bwd" in fresh_traffic:
                t = fresh_traffic["render_bwd"]
                traffic = (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0
                traffic_note = "measured in this run: " + fresh_note
            elif tpath and (N, H, B) == (200_000, 1024, 2):
                t = json.load(open(tpath)).get("render_bwd")
                if t:
                    traffic = (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0
                    traffic_note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of this command, "
                                    f"{os.path.relpath(tpath, ROOT)}; 2*FETCH+WRITE (gfx950 wide-read correction)")
            out["roofline_raster_bwd"] = {
                "kernel": "render_bwd", "bound": "hbm", "achieved": rb["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rb["frac_of_peak"], "traffic": traffic, "traffic_note": traffic_note, "avg_us": rb["avg_us"],
                "frames_per_launch": B, "algorithmic_bytes_per_launch": alg["render_bwd"],
                "measured": f"HIP events, {probe} fully instrumented warm-up steps"}
        fwd_us = sum(table[k]["us_per_iter"] for k in ("preprocess", "binning", "render_fwd") if k in table)
        bwd_us = sum(table[k]["us_per_iter"] for k in ("render_bwd", "preprocess_bwd") if k in table)
        # this rank's iteration by component (the terms of DESIGN.md section 6's scaling model), from the instrumented
        # warm-up steps: kernel families by HIP events, collectives by events around the calls, the rest by difference
        fam = lambda names: sum(kern[k]["us_per_iter"] for k in names if k in kern)
        dec_us = fam(("mlp_fwd", "mlp_stats", "layer_bwd", "mlp_bwd_data", "wgrad_act", "wgrad_reduce", "head_bwd", "bwd_stats"))
        coll = {k: {"us_per_iter": 1e3 * ms / probe, "calls_per_iter": n / probe} for k, (ms, n) in probe_c.items()}
        coll_us = sum(v["us_per_iter"] for v in coll.values())
        out["rank0_breakdown"] = {
            "measured": f"{probe} instrumented warm-up steps of rank 0 (every launch bracketed by HIP events: the step itself "
                        f"is slower than a timed one); frames per rank {B}",
            "step_us_instrumented": 1e3 * probe_ms, "decoder_us": dec_us, "raster_fwd_us": fwd_us, "raster_bwd_us": bwd_us,
            "loss_us": fam(("ssim_fwd", "ssim_bwd")), "collectives_us": coll_us, "collectives": coll,
            "other_us": 1e3 * probe_ms - dec_us - fwd_us - bwd_us - fam(("ssim_fwd", "ssim_bwd")) - coll_us,
            "other_is": "geometry net (conv5 kernels, up-sampling), LBS / skinning, pose encoder (stage 2), decode_pack, Adam, "
                        "launch gaps of the instrumented step"}
        out["kernels"] = {
            "measured": f"HIP events around every launch during the last {probe} warm-up steps (instrumenting all "
                        f"~100 launches costs ~1.4 ms/iteration, so the timed steps only carry the dominant family's)",
            "probe_pairs_per_frame": probe_pairs, "per_kernel": kern, "raster_per_stage": table, "raster_fwd_us": fwd_us, "raster_bwd_us": bwd_us,
            "raster_bwd_frac_of_8TBps": (alg["render_bwd"] + alg["preprocess_bwd"]) / (bwd_us * 1e-6) / 1e9 / HBM_PEAK_GBS if bwd_us else None}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, B)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
