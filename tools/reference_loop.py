#!/usr/bin/env python
"""Throughput of the REFERENCE'S OWN training loop on this repository.

    python tools/reference_loop.py --iters 200 [--out gpurun_out/x/reference_loop.json]

Writes a synthetic dataset in the reference's on-disk layout at the headline size (200k Gaussians on a 512^2 UV map,
1024 x 1024 frames, batch 2) and executes the reference's train.py — the byte-identical fixture
tests/golden/reference_scripts/train.py.txt — through gaussianavatar_amd.run_reference for `--iters` iterations. The
script is not edited: its per-iteration `loss.item()` (train.py:101), `backward(retain_graph=True)` (:95), its separate
`l1_loss_w` / `ssim` calls (:74-75), its DataLoader (4 workers decoding PNG frames) all run as written. Iteration
times are taken from outside: AvatarModel.step is wrapped to stamp the host clock (the loop's `.item()` has drained the
GPU by then). Printed next to it: the same iteration driven by bench.py's loop (no sync, frames resident on the device).
Third-party imports the image lacks (lpips, open3d, torchvision, torchmetrics) come from tests/stubs."""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--points", type=int, default=200_000)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--out", default="")
    ap.add_argument("--cache", type=int, default=-1, help="decoded-frame cache of the dataset reader in MB (-1: its default)")
    args = ap.parse_args()
    from gaussianavatar_amd.synthetic import make_assets, make_frames, write_dataset
    tmp = tempfile.mkdtemp(prefix="ga_refloop_")
    assets = make_assets(args.points, 512, "smpl")
    frames = make_frames(assets, args.frames, args.size, args.size)
    images = torch.ones(args.frames, 3, args.size, args.size)
    images[:, :, args.size // 5: 4 * args.size // 5, 2 * args.size // 5: 3 * args.size // 5] = 0.6
    t0 = time.perf_counter()
    paths = write_dataset(os.path.join(tmp, "data"), os.path.join(tmp, "proj"), assets, frames, images=images,
                          inp_posmap_size=128, stage2=False)
    t_write = time.perf_counter() - t0
    per_epoch = args.frames // 2
    epochs = (args.iters + per_epoch - 1) // per_epoch
    argv = ["-s", paths["source_path"], "-m", os.path.join(tmp, "out"), "--project_path", paths["project_path"],
            "--smpl_model_path", paths["smpl_model_path"], "--smplx_model_path", paths["smplx_model_path"],
            "--test_folder", paths["test_folder"], "--epochs", str(epochs), "--batch_size", "2",
            "--save_epoch", str(10 ** 6), "--save_epochs", "0", "--train_stage", "1", "--quiet"]
    from gaussianavatar_amd import run_reference
    import gaussianavatar_amd.avatar_model as AM
    if args.cache >= 0:
        import gaussianavatar_amd.dataset as D
        D._MonoBase.CACHE_MB = args.cache          # (the worker processes are forked later and inherit it)
    stamps = []
    orig_step = AM.AvatarModel.step
    # host time spent inside our entry points, per iteration (the remainder of an iteration is the script's own code,
    # its loss.item() wait included)
    parts = {k: [] for k in ("next_batch", "train_stage1", "zero_grad", "backward", "step")}

    def timed(name, fn):
        def w(*a, **kw):
            t = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                parts[name].append(time.perf_counter() - t)
        return w

    pair_windows = []                                  # mean (tile, Gaussian) pairs per frame (rasterizer.pair_statistics), 40 iterations at a time

    def step(self, epoch):
        t = time.perf_counter()
        r = orig_step(self, epoch)
        stamps.append(time.perf_counter())
        parts["step"].append(stamps[-1] - t)
        if len(stamps) % 40 == 0:                      # (blocks on the status records; the script's .item() is about to anyway)
            from gaussianavatar_amd import rasterizer
            pair_windows.append(round(rasterizer.pair_statistics(reset=True)[1]))
        return r

    orig = dict(stage1=AM.AvatarModel.train_stage1, zero=AM.AvatarModel.zero_grad, bwd=torch.Tensor.backward,
                it=AM._DeviceLoader.__iter__)
    AM.AvatarModel.step = step
    AM.AvatarModel.train_stage1 = timed("train_stage1", orig["stage1"])
    AM.AvatarModel.zero_grad = timed("zero_grad", orig["zero"])
    torch.Tensor.backward = timed("backward", orig["bwd"])

    def timed_iter(self):
        it = orig["it"](self)
        while True:
            t = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            parts["next_batch"].append(time.perf_counter() - t)
            yield b

    AM._DeviceLoader.__iter__ = timed_iter
    run_reference.install_paths()
    sys.path.append(os.path.join(ROOT, "tests", "stubs"))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        run_reference.run(os.path.join(ROOT, "tests", "golden", "reference_scripts", "train.py.txt"), argv)
    finally:
        os.chdir(cwd)
        AM.AvatarModel.step = orig_step
        AM.AvatarModel.train_stage1, AM.AvatarModel.zero_grad = orig["stage1"], orig["zero"]
        torch.Tensor.backward, AM._DeviceLoader.__iter__ = orig["bwd"], orig["it"]
    torch.cuda.synchronize()
    n = len(stamps)
    skip = min(40, n // 4)                               # warm-up: allocator, capacity history, worker start
    dt = [b - a for a, b in zip(stamps[skip:-1], stamps[skip + 1:])]
    dts = sorted(dt)
    res = {"script": "tests/golden/reference_scripts/train.py.txt (byte-identical to the reference's train.py)",
           "workload": f"stage 1, {args.points} Gaussians, {args.size}x{args.size}, batch 2, {args.frames} frames on disk (PNG), "
                       f"DataLoader with 4 workers as the reference configures it",
           "iterations": n, "timed": len(dt), "iters_per_s_mean": len(dt) / sum(dt),
           "ms_per_iter_median": 1e3 * dts[len(dts) // 2], "ms_per_iter_p10": 1e3 * dts[len(dts) // 10],
           "ms_per_iter_p90": 1e3 * dts[9 * len(dts) // 10], "dataset_write_s": t_write,
           "pairs_per_frame_mean_by_40_iterations": pair_windows,
           "host_ms_inside_our_entry_points_median": {k: 1e3 * sorted(v[skip:])[len(v[skip:]) // 2] for k, v in parts.items() if len(v) > skip}}
    ours = sum(res["host_ms_inside_our_entry_points_median"].values())
    res["host_ms_script_own_code_and_item_wait_median"] = res["ms_per_iter_median"] - ours
    print(json.dumps(res))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
