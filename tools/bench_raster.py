#!/usr/bin/env python
"""Rasterizer-only micro-benchmark: forward + backward of the HIP rasterizer alone, two frames per
launch as in the training iteration, on the synthetic body at 200k Gaussians / 1024^2 and 300k / 1920x1080.

Gaussian sets (seeded; positions = the synthetic avatar's skinned points, frames 0 and 1 of the synthetic dataset):
  avatar_3mm            section 8d "avatar-like": isotropic log-normal scales (median 3 mm, sigma_ln 0.4), opacity 1,
                        identity rotation, colour U(0,1)
  avatar_10mm / _20mm   the same with the median at 10 / 20 mm: the sizes a from-scratch training passes through during
                        the reference's scale warm-up (model/avatar_model.py:315-316) -> ~2 M / ~4 M pairs per frame
  general               section 8d "general": anisotropic log-normal scales (median 3 mm, sigma_ln 0.7 per axis), random
                        unit quaternions, opacity U(0.05, 1)
  general_sh3           the general set with SH colours of degree 3 (one frame per launch: the single-frame entry point)
  warmup_K              what bench.py --iteration K renders: the model trained for 12 iterations from its initial state with
                        the reference's scale warm-up at K / 1000, then frozen (scales AND positions from the net)
Per set and seed: 10 warm-up + ITERS timed iterations, every kernel bracketed with HIP events on its stream (the
library's own profiler, include/gsr.h gsr_profile_*); reported: the MEDIAN over all timed iterations of all seeds, per
kernel, in us per launch (2 frames), the pair count, and the algorithmic GB/s of each kernel (bench.py:
algorithmic_bytes). --pmc adds HBM-side traffic per launch from two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: separate
passes, MI355X_MICROARCH.md; traffic = 2 x FETCH + WRITE, values in KiB) of a child run of ONE set.

  python tools/bench_raster.py --out profiles/r06_raster_ubench.json [--pmc] [--sets a,b] [--sizes 200k,300k]
"""
import argparse, glob, json, math, os, shutil, statistics, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import algorithmic_bytes
from gaussianavatar_amd import rasterizer
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
from gaussianavatar_amd.lbs import skin
from gaussianavatar_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, PROFILE_KERNELS,
                                           rasterize_gaussians_batch)

SIZES = {"200k": (200000, 1024, 1024), "300k": (300000, 1920, 1080)}
ALL_SETS = ("avatar_3mm", "avatar_10mm", "avatar_20mm", "warmup_150", "warmup_300", "general", "general_sh3")
BINNING = ("tile_scan", "scatter", "tile_sort")


def body(N, W, H, frames=2):
    """Skinned positions [frames, N, 3] of the synthetic avatar and the frames' cameras."""
    mp, npar, op = default_params(batch_size=frames, num_points=N, image_width=W, image_height=H,
                                  query_posmap_size=1024 if N > 512 * 512 else 512)
    m = AvatarModel(mp, npar, op, train=True)
    bt = collate_frames([m.train_dataset[i] for i in range(frames)], "cuda")
    with torch.no_grad():
        live = m._body(m.pose.weight[:frames], m.transl.weight[:frames], None)
        pts = skin(m.query_points[0], None, m.query_lbs[0], live.cano2live).contiguous()
    return m, bt, pts


def warmup_set(K, N, W, H, seed, steps=12):
    """The scene a from-scratch training renders at iteration K of the reference's scale warm-up: the model of bench.py
    trained for `steps` iterations with every Gaussian scaled by K / 1000 (bench.py --iteration K), then frozen.
    Returns (positions [2,N,3], Gaussian set)."""
    from gaussianavatar_amd.losses import l1_loss_w, ssim, weighted_sum
    torch.manual_seed(seed)
    mp, npar, op = default_params(batch_size=2, num_points=N, image_width=W, image_height=H, num_frames=16,
                                  query_posmap_size=1024 if N > 512 * 512 else 512)
    m = AvatarModel(mp, npar, op, train=True)
    m.training_setup(); m.net.train()
    gt = torch.ones(2, 3, H, W, device="cuda")
    gt[:, :, H // 5: 4 * H // 5, 2 * W // 5: 3 * W // 5] = 0.6
    ds = m.train_dataset
    batches = [collate_frames([ds[(s * 2 + k) % len(ds)] for k in range(2)], "cuda") for s in range(4)]
    l = op.lambda_dssim
    dbg = bool(os.environ.get("RUB_DEBUG"))
    for i in range(steps):
        if dbg:
            print(f"  warmup_set step {i} capacity {rasterizer._capacity.capacity((N, W, H))} last {rasterizer.last_status()}", flush=True)
        image, _p, off, geo, scl = m.train_stage1(batches[i % 4], K)
        loss = weighted_sum([scl, off, l1_loss_w(image, gt), ssim(image, gt), geo],
                            [op.lambda_scale, op.lambda_rgl, 1.0 - l, -l, 1.0], bias=l)
        m.zero_grad(1); loss.backward(); m.step(1)
    bt = batches[0]
    with torch.no_grad():
        live = m._body(m.pose(bt["pose_idx"]), m.transl(bt["pose_idx"]), None)
        _o, _s, res, scales, colors = m._decode(2, None, K, True)
        pts = skin(m.query_points[0], res, m.query_lbs[0], live.cano2live).contiguous()
    rots = m.fix_rotation if m.fix_rotation.dim() == 2 else m.fix_rotation[0]
    gs = dict(colors=colors[0].contiguous(), scales=scales[0].contiguous(), rots=rots.contiguous(),
              opac=m.fix_opacity.reshape(-1, 1).contiguous(), shs=None)
    return m, bt, pts, gs


def gaussian_set(kind, N, seed, dev="cuda"):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)
    colors = ru(N, 3)
    if kind.startswith("avatar_"):
        med = float(kind.split("_")[1].rstrip("m")) * 1e-3
        scales = torch.exp(rn(N, 1) * 0.4 + math.log(med)).repeat(1, 3)
        rots = torch.zeros(N, 4); rots[:, 0] = 1
        opac = torch.ones(N, 1)
    else:
        scales = torch.exp(rn(N, 3) * 0.7 + math.log(3e-3))
        rots = torch.nn.functional.normalize(rn(N, 4), dim=1)
        opac = ru(N, 1) * 0.95 + 0.05
    shs = (rn(N, 16, 3) * 0.3) if kind.endswith("sh3") else None
    to = lambda t: None if t is None else t.to(dev).contiguous()
    return dict(colors=to(colors), scales=to(scales), rots=to(rots), opac=to(opac), shs=to(shs))


def settings(m, bt, W, H, frames, sh_degree=0):
    sel = (lambda t: t[:frames]) if frames > 1 else (lambda t: t[0])
    return GaussianRasterizationSettings(H, W, math.tan(float(bt["FovX"][0]) * 0.5), math.tan(float(bt["FovY"][0]) * 0.5),
                                         m.background, 1.0, sel(bt["world_view_transform"]), sel(bt["full_proj_transform"]),
                                         sh_degree, bt["camera_center"][0], False, False)


def make_iteration(kind, m, bt, pts, gs, W, H):
    """Returns (callable running one forward + backward, frames per launch)."""
    if gs["shs"] is None:
        rs = settings(m, bt, W, H, 2)
        gout = torch.randn(2, 3, H, W, device="cuda")
        leaves = [pts.clone().requires_grad_(True), gs["colors"].clone().requires_grad_(True),
                  gs["opac"].clone().requires_grad_(True), gs["scales"].clone().requires_grad_(True),
                  gs["rots"].clone().requires_grad_(True)]

        def it():
            img, _ = rasterize_gaussians_batch(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], rs)
            img.backward(gout)
            for t in leaves:
                t.grad = None
        return it, 2
    rs = settings(m, bt, W, H, 1, sh_degree=3)
    gout = torch.randn(3, H, W, device="cuda")
    leaves = [pts[0].clone().requires_grad_(True), gs["shs"].clone().requires_grad_(True),
              gs["opac"].clone().requires_grad_(True), gs["scales"].clone().requires_grad_(True),
              gs["rots"].clone().requires_grad_(True)]
    rast = GaussianRasterizer(rs)

    def it1():
        img, _ = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), opacities=leaves[2], shs=leaves[1],
                      scales=leaves[3], rotations=leaves[4])
        img.backward(gout)
        for t in leaves:
            t.grad = None
    return it1, 1


def survivor_records(m, bt, pts, gs, W, H):
    """(pairs, recorded segments, survivor records = atomic gradient records of one backward pass, occupied tiles,
    longest list) of frame 0 — from the state the forward pass leaves."""
    if gs["shs"] is not None:
        return None
    rs = settings(m, bt, W, H, 1)
    args = (rs, pts[0], gs["colors"], gs["opac"], gs["scales"], gs["rots"])
    _c, _r, v, status = rasterizer.rasterize_with_state(*args)
    if status[1]:      # the default capacity overflowed (this entry point does not retry): render again with room
        del v
        _c, _r, v, status = rasterizer.rasterize_with_state(*args, max_pairs=int(status[0]) + 4096)
    sc = v["seg_count"].long()
    T, Bk = sc.shape
    off = v["tile_offset"].long()
    start, end = off[:-1], off[1:]
    first = Bk * ((start >> 6) + torch.arange(T, device=start.device))
    cap = (end >> 6) - (start >> 6) + 1
    base = (first[:, None] + torch.arange(Bk, device=start.device)[None, :] * cap[:, None]).reshape(-1)
    cnt = sc.reshape(-1)
    nz = cnt > 0
    base, cnt = base[nz], cnt[nz]
    seg_first = torch.repeat_interleave(base, cnt)
    within = torch.arange(int(cnt.sum()), device=cnt.device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
    entries = int(v["seg_info"][(seg_first + within), 1].long().sum())
    n = (end - start)
    tt = v["tiles_touched"].float()
    q = torch.quantile(tt[tt > 0], torch.tensor([0.5, 0.9, 0.99, 0.999], device=tt.device)).tolist()
    return dict(pairs=int(status[0]), segments=int(cnt.sum()), records=entries, occupied_tiles=int((n > 0).sum()),
                longest_list=int(n.max()), lists_over_8192=int((n > 8192).sum()), lists_over_2048=int((n > 2048).sum()),
                tiles_per_gaussian_p50_p90_p99_p999=[round(x, 1) for x in q], tiles_per_gaussian_max=int(tt.max()),
                gaussians_over_64_tiles=int((tt > 64).sum()), gaussians_over_1024_tiles=int((tt > 1024).sum()))


def run_set(kind, size_key, seeds, iters, scene, warm=10):
    N, W, H = SIZES[size_key]
    m, bt, pts = scene
    per_kernel = {k: [] for k in PROFILE_KERNELS}
    walls, pairs_l = [], []
    rec = None
    frames = 2
    for seed in seeds:
        if kind.startswith("warmup_"):
            if os.environ.get("RUB_TRACE") == kind:
                from gaussianavatar_amd import _native
                _native.gsr().gsr_set_trace(1)
            m, bt, pts, gs = warmup_set(int(kind.split("_")[1]), N, W, H, seed)
            if os.environ.get("RUB_DEBUG"):
                torch.cuda.synchronize(); print("  warmup_set done", flush=True)
        else:
            gs = gaussian_set(kind, N, seed)
        if rec is None:
            rec = survivor_records(m, bt, pts, gs, W, H)
        it, frames = make_iteration(kind, m, bt, pts, gs, W, H)
        rasterizer.profile_enable(False)
        for _ in range(warm):
            it()
        rasterizer.check_overflow(True); rasterizer.pair_statistics(reset=True)
        rasterizer.profile_enable(True); rasterizer.profile_read(True)
        for _ in range(iters):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            it()
            torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e6)
            for k, (ms, c) in rasterizer.profile_read(True).items():
                if c:
                    per_kernel[k].append(ms * 1e3)
        _n, p = rasterizer.pair_statistics(True)
        pairs_l.append(p)
        rasterizer.profile_enable(False)
    # the forward-only render of the last seed's scene (torch.no_grad(): eval.py / render_novel_pose.py), 2 frames per launch
    eval_us = None
    if gs["shs"] is None:
        rs2 = settings(m, bt, W, H, 2)
        def ev():
            with torch.no_grad():
                rasterize_gaussians_batch(pts, gs["colors"], gs["opac"], gs["scales"], gs["rots"], rs2)
        for _ in range(5):
            ev()
        rasterizer.check_overflow(True)
        rasterizer.profile_enable(["render_fwd"]); rasterizer.profile_read(True)
        for _ in range(20):
            ev()
        ms, c = rasterizer.profile_read(True)["render_fwd"]
        rasterizer.profile_enable(False)
        eval_us = round(ms / c * 1e3, 1) if c else None
    med = {k: statistics.median(v) for k, v in per_kernel.items() if v}
    D = statistics.mean(pairs_l)                        # pairs per frame (the capacity poll's mean over the launches)
    ab = algorithmic_bytes(N, D, W * H)
    med["binning"] = sum(med.get(k, 0.0) for k in BINNING)
    gbs = {k: frames * ab[k] / (med[k] * 1e-6) / 1e9 for k in ab if med.get(k)}
    out = dict(set=kind, size=size_key, gaussians=N, image=[H, W], frames_per_launch=frames, seeds=list(seeds),
               timed_iterations=iters * len(seeds), pairs_per_frame=D,
               us_per_launch_median={k: round(v, 1) for k, v in med.items()},
               wall_us_fwd_bwd_median=round(statistics.median(walls), 1), forward_only_render_us=eval_us,
               algorithmic_GBps={k: round(v, 1) for k, v in gbs.items()},
               frac_of_8TBps={k: round(v / 8000.0, 4) for k, v in gbs.items()},
               ps_per_pair={k: round(med[k] * 1e6 / (frames * D), 1) for k in med if D > 0},
               frame0=rec)
    return out


_FAMILIES = {"preprocess_kernel": "preprocess", "tile_scan_kernel": "tile_scan", "scatter_kernel": "scatter",
             "tile_sort_chunk_kernel": "tile_sort_chunk", "tile_merge": "tile_merge", "render_fwd_kernel": "render_fwd",
             "render_bwd_kernel": "render_bwd", "preprocess_bwd_kernel": "preprocess_bwd"}


def pmc_traffic(kind, size_key):
    """{kernel: {fetch_kb, write_kb, traffic_bytes}} per launch from two rocprofv3 --pmc passes of a child run."""
    import csv
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="ga_rub_", dir="/tmp")
    try:
        for counter, key in (("FETCH_SIZE", "fetch_kb"), ("WRITE_SIZE", "write_kb")):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--child", kind, size_key]
            try:
                rc = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, timeout=240).returncode
            except (OSError, subprocess.TimeoutExpired):
                return None
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None
            vals = {}
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r.get("Counter_Name") != counter:
                        continue
                    for sub, fam in _FAMILIES.items():
                        if sub in r["Kernel_Name"]:
                            vals.setdefault(fam, []).append(float(r["Counter_Value"]))
                            break
            for fam, v in vals.items():
                h = v[len(v) // 2:]
                out.setdefault(fam, {})[key] = sum(h) / len(h)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for fam, v in out.items():
        if "fetch_kb" in v and "write_kb" in v:
            v["traffic_bytes"] = (2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0
    return out


def child(kind, size_key):
    N, W, H = SIZES[size_key]
    if kind.startswith("warmup_"):
        m, bt, pts, gs = warmup_set(int(kind.split("_")[1]), N, W, H, 0)
    else:
        m, bt, pts = body(N, W, H)
        gs = gaussian_set(kind, N, 0)
    it, _ = make_iteration(kind, m, bt, pts, gs, W, H)
    for _ in range(12):
        it()
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--sets", default=",".join(ALL_SETS))
    ap.add_argument("--sizes", default="200k,300k")
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--pmc", action="store_true", help="add PMC traffic for avatar_3mm, avatar_20mm and general")
    ap.add_argument("--child", nargs=2, default=None)
    a = ap.parse_args()
    if a.child:
        return child(*a.child)
    torch.manual_seed(0)
    rows = []
    for size_key in a.sizes.split(","):
        scene = body(*SIZES[size_key])
        for kind in a.sets.split(","):
            seeds = range(1) if (kind.endswith("sh3") or kind.startswith("warmup_")) else range(a.seeds)
            r = run_set(kind, size_key, seeds, a.iters, scene)
            if a.pmc and kind in ("avatar_3mm", "avatar_20mm", "general", "warmup_300"):
                torch.cuda.synchronize()
                tr = pmc_traffic(kind, size_key)
                r["pmc_traffic_per_launch"] = tr
                if tr:
                    ab = algorithmic_bytes(r["gaussians"], r["pairs_per_frame"], r["image"][0] * r["image"][1])
                    r["traffic_over_algorithmic"] = {
                        k: round(tr[k]["traffic_bytes"] / (r["frames_per_launch"] * ab[k]), 2)
                        for k in ("preprocess", "render_fwd", "render_bwd", "preprocess_bwd") if k in tr and "traffic_bytes" in tr[k]}
            rows.append(r)
            u = r["us_per_launch_median"]
            print(f"{size_key} {kind:12s} pairs/frame {r['pairs_per_frame']:9.0f}  " +
                  "  ".join(f"{k}={u[k]:.0f}" for k in list(PROFILE_KERNELS) + ["binning"] if k in u) +
                  f"  wall={r['wall_us_fwd_bwd_median']:.0f}us  eval_render={r['forward_only_render_us']}  records/pair="
                  f"{(r['frame0']['records'] / max(r['frame0']['pairs'], 1)) if r['frame0'] else float('nan'):.2f}", flush=True)
    doc = {"tool": "tools/bench_raster.py", "unit": "us per launch (median over seeds x iterations), 2 frames per launch "
           "unless frames_per_launch says otherwise", "device": torch.cuda.get_device_name(0), "rows": rows}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
