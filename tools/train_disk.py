"""End-to-end run on a dataset in the reference's on-disk layout.

    python tools/train_disk.py --out /tmp/ga_run --points 20000 --size 256 --frames 8 --epochs 30

1. a "teacher" avatar (seeded synthetic assets, net at a different random init) renders the
   training frames; images, masks, cameras, SMPL parameters, posmaps, lbs map, uv mask and an
   SMPL-shaped body-model file are written in the reference's formats (synthetic.write_dataset);
2. the training loop of /root/reference/train.py:31-135 (stage 1: L1 + SSIM + regularisers,
   zero_grad / backward(retain_graph=True) / step, save every `save_epoch`) runs on
   AvatarModel(source_path=...) — the data comes back through MonoDataset_train;
3. the loops of /root/reference/eval.py:39-78 and render_novel_pose.py:12-38 (default collate,
   batch_size 1, to_cuda, `image, = render_free_stage1(batch, 59400)`) render the test split, the
   novel poses and an orbit of novel views from the saved checkpoint; PSNR against the teacher's
   frames is printed (lpips / torchmetrics / open3d logging of the reference scripts are left out:
   not installed, off the scoped path).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params  # noqa: E402
from gaussianavatar_amd.dataset import to_cuda  # noqa: E402
from gaussianavatar_amd.losses import adjust_loss_weights, l1_loss_w, ssim, weighted_sum  # noqa: E402
from gaussianavatar_amd.synthetic import make_assets, make_frames, write_dataset  # noqa: E402


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a - b) ** 2).clamp_min(1e-12)))


def make_teacher_dataset(out, points, size, frames_n, uv, inp, smpl_type="smpl", device="cuda"):
    assets = make_assets(points, uv, smpl_type)
    frames = make_frames(assets, frames_n, size, size)
    mp, npar, op = default_params(query_posmap_size=uv, inp_posmap_size=inp, smpl_type=smpl_type, batch_size=1)
    torch.manual_seed(1234)
    teacher = AvatarModel(mp, npar, op, assets=assets, frames=frames, train=False, device=device)
    teacher.net.train()                      # BatchNorm with batch statistics, as in training
    with torch.no_grad():
        teacher.geo_feature.normal_(0.0, 0.3)
        images = []
        for i in range(frames_n):
            batch = collate_frames([teacher._free_dataset()[i]], device)
            images.append(teacher.render_free_stage1(batch, 59400)[0].clamp(0, 1).cpu())
    images = torch.stack(images)
    masks = (images < 0.999).any(dim=1)
    paths = write_dataset(os.path.join(out, "data"), os.path.join(out, "project"), assets, frames,
                          images=images, masks=masks, inp_posmap_size=inp, stage2=False)
    return paths, images


def train(model, net, opt, log=print):
    """train.py:31-135, stage 1."""
    avatarmodel = AvatarModel(model, net, opt, train=True)
    train_loader = avatarmodel.getTrainDataloader()
    first_iter, epoch_start = 0, 0
    avatarmodel.training_setup()
    history = []
    t0 = time.time()
    for epoch in range(epoch_start + 1, opt.epochs + 1):
        avatarmodel.net.train()
        avatarmodel.pose.train()
        avatarmodel.transl.train()
        wdecay_rgl = adjust_loss_weights(opt.lambda_rgl, epoch, mode="decay", start=epoch_start, every=20)
        for _, batch_data in enumerate(train_loader):
            first_iter += 1
            batch_data = to_cuda(batch_data, device=avatarmodel.device)
            gt_image = batch_data["original_image"]
            image, points, offset_loss, geo_loss, scale_loss = avatarmodel.train_stage1(batch_data, first_iter)
            # train.py:70-77: lambda_scale*scale + wdecay_rgl*offset + (1-l)*L1 + l*(1 - SSIM) + geo
            l = opt.lambda_dssim
            loss = weighted_sum([scale_loss, offset_loss, l1_loss_w(image, gt_image), ssim(image, gt_image), geo_loss],
                                [opt.lambda_scale, wdecay_rgl, 1.0 - l, -l, 1.0], bias=l)
            avatarmodel.zero_grad(epoch)
            loss.backward(retain_graph=True)
            avatarmodel.step(epoch)
            history.append(float(loss.detach()))
        if epoch % model.save_epoch == 0 or epoch == opt.epochs:
            avatarmodel.save(epoch)
        if epoch % max(1, opt.epochs // 5) == 0:
            log("epoch %d  loss %.5f  (%.1f it/s)" % (epoch, history[-1], first_iter / (time.time() - t0)))
    return avatarmodel, history


def render_sets(model, net, opt, epoch, which="test"):
    """eval.py:39-78 / render_novel_pose.py:12-38: returns the rendered images (and the ground
    truth where the split has one)."""
    out, gts = [], []
    with torch.no_grad():
        avatarmodel = AvatarModel(model, net, opt, train=False)
        avatarmodel.training_setup()
        avatarmodel.load(epoch, test=(which != "test"))
        avatarmodel.net.train()
        if which == "test":
            ds = avatarmodel.getTestDataset()
        elif which == "novel_pose":
            ds = avatarmodel.getNovelposeDataset()
        else:
            ds = avatarmodel.getNovelviewDataset()
            ds.update_smpl(0, 6)
        loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
        for _idx, batch_data in enumerate(loader):
            batch_data = to_cuda(batch_data, device=avatarmodel.device)
            image, = avatarmodel.render_free_stage1(batch_data, 59400)
            out.append(image)
            if "original_image" in batch_data:
                gts.append(batch_data["original_image"][0])
    return out, gts


def run(out, points=20000, size=256, frames=8, epochs=30, uv=256, inp=64, log=print):
    paths, teacher_images = make_teacher_dataset(out, points, size, frames, uv, inp)
    model, net, opt = default_params(query_posmap_size=uv, inp_posmap_size=inp, epochs=epochs,
                                     model_path=os.path.join(out, "output"), save_epoch=max(1, epochs // 2), **paths)
    opt.sched_milestones = [int(epochs / 3), int(epochs * 2 / 3)]
    torch.manual_seed(0)
    untrained, gts = render_sets_untrained(model, net, opt)
    avatarmodel, history = train(model, net, opt, log)
    test_imgs, gts = render_sets(model, net, opt, epochs, "test")
    novel, _ = render_sets(model, net, opt, epochs, "novel_pose")
    orbit, _ = render_sets(model, net, opt, epochs, "novel_view")
    res = dict(loss_first=float(np.mean(history[:3])), loss_last=float(np.mean(history[-3:])),
               psnr_untrained=float(np.mean([psnr(a, b) for a, b in zip(untrained, gts)])),
               psnr_test=float(np.mean([psnr(a, b) for a, b in zip(test_imgs, gts)])),
               novel_pose_frames=len(novel), novel_pose_shape=list(novel[0].shape),
               novel_view_frames=len(orbit), iterations=len(history))
    return res


def render_sets_untrained(model, net, opt):
    with torch.no_grad():
        m = AvatarModel(model, net, opt, train=False)
        m.net.train()
        ds = m.getTestDataset()
        out, gts = [], []
        for batch_data in torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0):
            batch_data = to_cuda(batch_data, device=m.device)
            image, = m.render_free_stage1(batch_data, 59400)
            out.append(image)
            gts.append(batch_data["original_image"][0])
    return out, gts


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/ga_run")
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--uv", type=int, default=256)
    ap.add_argument("--inp", type=int, default=64)
    a = ap.parse_args()
    print(json.dumps(run(a.out, a.points, a.size, a.frames, a.epochs, a.uv, a.inp)))
