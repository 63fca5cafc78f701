"""GPU integration: the AvatarModel hot path end to end, checked against an all-CPU evaluation
of the reference's formulas (torch CPU nets + oracle LBS + oracle rasterizer)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small_model(stage=1, smpl_type="smpl", B=2, N=3000, size=96):
    from gaussianavatar_amd.avatar_model import AvatarModel, default_params
    torch.manual_seed(0)
    mp, npar, op = default_params(batch_size=B, num_points=N, query_posmap_size=64, inp_posmap_size=32,
                                  image_width=size, image_height=size, num_frames=4, train_stage=stage,
                                  smpl_type=smpl_type, c_geom=16, c_pose=16, hsize=32, nf=4)
    m = AvatarModel(mp, npar, op, train=True)
    m.training_setup()
    return m, mp, npar, op


def test_stage1_image_matches_cpu_reference_formulas(raster_oracle):
    from gaussianavatar_amd.avatar_model import collate_frames
    from oracle import lbs_oracle as O
    from tests.scenes import cam_kwargs
    import math
    m, mp, npar, op = _small_model()
    batch = collate_frames([m.train_dataset[i] for i in range(2)], "cuda")
    it = 40
    with torch.no_grad():
        image, full_pred, *_ = m.train_stage1(batch, it)
    # --- CPU evaluation with the reference's own layout: net on B expanded copies, permutes, mask
    import copy
    net = copy.deepcopy(m.net).cpu()
    net.train()
    B = 2
    geom = m.geo_feature.detach().cpu().expand(B, -1, -1, -1).contiguous()
    uv = m.uv_coord_map.cpu()[None].expand(B, -1, -1).contiguous()
    res, sc, shs = net(None, geom, uv)
    res = res.permute(0, 2, 1) * 0.02
    valid = m.valid_idx.cpu()
    point_res = res[:, valid]
    A = O.joint_transforms(m.pose.weight.detach().cpu()[:2], m.transl.weight.detach().cpu()[:2],
                           m.smpl_model.joints_rest.cpu(), m.smpl_model.parents.long().cpu())
    M = A @ m.inv_mats.cpu()[:B]
    full = O.skin(m.query_points.cpu(), point_res, m.query_lbs.cpu(), M)
    np.testing.assert_allclose(full_pred.cpu().numpy(), full.detach().numpy(), atol=5e-5)
    scales = (sc.permute(0, 2, 1) * 1e-3 * it)[:, valid].repeat(1, 1, 3)
    cols = shs.permute(0, 2, 1)[:, valid]
    cam = m.frames["camera"]
    for b in range(B):
        ref = raster_oracle.forward(
            full[b].detach().numpy(), cols[b].detach().numpy(), np.ones(full.shape[1], np.float32),
            scales[b].detach().numpy(), m.fix_rotation.cpu().numpy(),
            viewmatrix=cam["world_view_transform"], projmatrix=cam["full_proj_transform"],
            bg=np.ones(3, np.float32), W=cam["width"], H=cam["height"],
            tanfovx=math.tan(cam["FovX"] * 0.5), tanfovy=math.tan(cam["FovY"] * 0.5))
        d = np.abs(image[b].cpu().numpy() - ref["color"])
        assert d.mean() <= 1e-4, d.mean()


def test_stage1_training_reduces_loss_and_checkpoint_roundtrip(tmp_path):
    from gaussianavatar_amd.avatar_model import collate_frames
    from gaussianavatar_amd.losses import l1_loss_w, ssim
    m, mp, npar, op = _small_model()
    m.model_path = str(tmp_path)
    batch = collate_frames([m.train_dataset[i] for i in range(2)], "cuda")
    gt = torch.ones(2, 3, 96, 96, device="cuda")
    gt[:, :, 30:70, 40:56] = 0.2
    losses = []
    for it in range(1, 31):
        image, pts, offset_loss, geo_loss, scale_loss = m.train_stage1(batch, 100 + it)
        loss = 0.8 * l1_loss_w(image, gt) + 0.2 * (1 - ssim(image, gt)) + 10 * offset_loss + geo_loss + 0.03 * scale_loss
        m.zero_grad(1)
        loss.backward(retain_graph=True)      # the reference's loop passes retain_graph=True
        m.step(1)
        losses.append(float(loss))
    assert np.isfinite(losses).all()
    assert losses[-1] < losses[0], losses
    m.save(7)
    with torch.no_grad():
        ref_img = m.render_free_stage1(dict(batch, pose_data=m.pose.weight[:2].detach(),
                                            transl_data=m.transl.weight[:2].detach()), 59400)
    m2, *_ = _small_model()
    m2.model_path = str(tmp_path)
    m2.load(7)
    with torch.no_grad():
        img2 = m2.render_free_stage1(dict(batch, pose_data=m.pose.weight[:2].detach(),
                                          transl_data=m.transl.weight[:2].detach()), 59400)
    assert float((ref_img - img2).abs().max()) < 1e-5


def test_pose_gradients_flow_to_embeddings():
    from gaussianavatar_amd.avatar_model import collate_frames
    m, *_ = _small_model()
    batch = collate_frames([m.train_dataset[i] for i in (1, 3)], "cuda")
    image, *_ = m.train_stage1(batch, 500)
    image.mean().backward()
    g = m.pose.weight.grad
    assert g.is_sparse
    dense = g.to_dense()
    assert dense[[1, 3]].abs().sum() > 0 and dense[[0, 2]].abs().sum() == 0
    assert m.transl.weight.grad.to_dense()[[1, 3]].abs().sum() > 0


@pytest.mark.parametrize("smpl_type", ["smpl", "smplx"])
def test_stage2_image_and_gradients_match_cpu_reference_formulas(smpl_type, raster_oracle):
    """train_stage2 (/root/reference/model/avatar_model.py:369-463: pose-encoder UNet, decoder per frame with
    BatchNorm statistics over the batch, no scale warm-up) on SMPL and on the SMPL-X-shaped body: image,
    pose_encoder / decoder / geometry-feature gradients and the sparse pose gradient against the all-CPU
    evaluation of the reference's formulas (tests/cpu_reference.py). Production widths (fused MFMA decoder)."""
    from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
    from tests import cpu_reference
    torch.manual_seed(0)
    mp, npar, op = default_params(batch_size=2, num_points=3000, query_posmap_size=64, inp_posmap_size=64,
                                  image_width=96, image_height=80, num_frames=4, train_stage=2, smpl_type=smpl_type)
    m = AvatarModel(mp, npar, op, train=True)
    m.training_setup()
    with torch.no_grad():                       # stand-in for the stage-1 checkpoint: ~1 cm Gaussians
        m.net.decoder.conv8N.bias.fill_(-4.0)
    batch = collate_frames([m.train_dataset[i] for i in (1, 2)], "cuda")
    snap = cpu_reference.snapshot(m)
    image, pts, pose_loss, offset_loss = m.train_stage2(batch, 1)
    w = torch.linspace(0.5, 1.5, image.numel(), device="cuda").reshape(image.shape)
    loss = ((1 - image) * w).mean() + 10 * pose_loss + 10 * offset_loss
    m.zero_grad(1)
    loss.backward()
    ref = cpu_reference.forward(m, snap, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()},
                                1, raster_oracle, stage=2)
    assert float((image.detach().cpu() - ref["image"].detach()).abs().mean()) <= 1e-4
    np.testing.assert_allclose(pts.detach().cpu().numpy(), ref["full_pred"].detach().numpy(), atol=5e-5)
    (((1 - ref["image"]) * w.cpu()).mean() + 10 * ref["pose_loss"] + 10 * ref["offset_loss"]).backward()
    pairs = [("net." + k, p.grad, dict(snap["net"].named_parameters())[k].grad) for k, p in m.net.named_parameters()]
    pairs += [("enc." + k, p.grad, dict(snap["pose_encoder"].named_parameters())[k].grad)
              for k, p in m.pose_encoder.named_parameters()]
    pairs.append(("geo", m.geo_feature.grad, snap["geo"].grad))
    pairs.append(("pose", m.pose.weight.grad.to_dense(), snap["pose"].grad))
    pairs.append(("transl", m.transl.weight.grad.to_dense(), snap["transl"].grad))
    gmax = max(float(c.abs().max()) for _, _, c in pairs if c is not None)
    for name, g, c in pairs:
        assert (g is None) == (c is None), name
        if c is None:
            continue
        err = float((g.cpu() - c).abs().max())
        assert err <= 5e-3 * float(c.abs().max()) + 2e-4 * gmax, (name, err, float(c.abs().max()), gmax)
    m.step(1)
    assert image.shape == (2, 3, 80, 96) and torch.isfinite(loss)


def test_stage1_on_a_uv_map_whose_texel_count_is_not_a_multiple_of_32(raster_oracle, monkeypatch):
    """A 60 x 60 query map (3,600 decoder rows: 3,600 % 32 = 16) at the production widths (c_geom 64, hsize 128): the
    one-call decoder and the one-pass layer backward need whole 32-row slabs, so THIS configuration runs on the
    per-launch path with the separate weight- / data-gradient kernels (csrc/ganet_mlp_bwd.hip, ganet_wgrad_split.hip) —
    the reason those kernels stay in the library. Image, regularisers and every gradient against the
    all-CPU evaluation of the reference's formulas (tests/cpu_reference.py)."""
    from gaussianavatar_amd import fused
    from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
    from tests import cpu_reference
    from tests.grad_check import assert_grads_close
    torch.manual_seed(0)
    mp, npar, op = default_params(batch_size=2, num_points=2500, query_posmap_size=60, inp_posmap_size=64,
                                  image_width=96, image_height=96, num_frames=4, train_stage=1)
    m = AvatarModel(mp, npar, op, train=True)
    m.training_setup()
    used = {"wgrad": 0, "bwd_data": 0, "fused": 0}
    lib = fused._native.ganet()
    for name, key in (("ganet_wgrad_act", "wgrad"), ("ganet_mlp_bwd_data", "bwd_data"), ("ganet_mlp_bwd_fused", "fused")):
        real = getattr(lib, name)
        def counted(*a, _real=real, _key=key):
            used[_key] += 1
            return _real(*a)
        monkeypatch.setattr(lib, name, counted)
    batch = collate_frames([m.train_dataset[i] for i in (0, 2)], "cuda")
    snap = cpu_reference.snapshot(m)
    image, pts, offset_loss, geo_loss, scale_loss = m.train_stage1(batch, 300)
    w = torch.linspace(0.5, 1.5, image.numel(), device="cuda").reshape(image.shape)
    loss = ((1 - image) * w).mean() + 10 * offset_loss + geo_loss + 0.03 * scale_loss
    m.zero_grad(1)
    loss.backward()
    assert used["wgrad"] >= 12 and used["bwd_data"] >= 10 and used["fused"] == 0, used
    ref = cpu_reference.forward(m, snap, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()},
                                300, raster_oracle, stage=1)
    assert float((image.detach().cpu() - ref["image"].detach()).abs().mean()) <= 1e-4
    (((1 - ref["image"]) * w.cpu()).mean() + 10 * ref["offset_loss"] + ref["geo_loss"] + 0.03 * ref["scale_loss"]).backward()
    pairs = [("net." + k, p.grad, dict(snap["net"].named_parameters())[k].grad) for k, p in m.net.named_parameters()]
    pairs.append(("geo", m.geo_feature.grad, snap["geo"].grad))
    assert_grads_close(pairs, rel=5e-3, floor=2e-4, cos_tol=1e-4)


def test_render_free_stage2_uses_the_learned_pose_embeddings():
    """/root/reference/model/avatar_model.py:555-560: stage-2 evaluation looks the pose up by `pose_idx`."""
    from gaussianavatar_amd.avatar_model import collate_frames
    m, *_ = _small_model(stage=2)
    ds = m.getTestDataset()
    batch = collate_frames([ds[1]], "cuda")
    with torch.no_grad():
        a = m.render_free_stage2(batch, 59400)
        m.pose.weight[1, 3:6] += 0.4                         # the embedding row, not batch['pose_data']
        b = m.render_free_stage2(batch, 59400)
        c = m.render_free_stage2(dict(batch, pose_data=batch["pose_data"] + 1.0), 59400)
    assert float((a - b).abs().max()) > 1e-3 and float((b - c).abs().max()) == 0.0


def test_train_eval_loops_on_disk_dataset(tmp_path):
    """tools/train_disk.py: a teacher avatar's frames written in the reference's on-disk layout,
    fitted through MonoDataset_train with the loop of the reference's train.py, then rendered from
    the checkpoint with the loops of eval.py / render_novel_pose.py (test split, novel poses,
    novel-view orbit)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "train_disk", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "train_disk.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(str(tmp_path), points=6000, size=128, frames=4, epochs=300, uv=128, inp=32, log=lambda *_: None)
    assert res["iterations"] == 600 and res["novel_pose_frames"] == 4 and res["novel_view_frames"] == 6
    assert res["novel_pose_shape"] == [3, 1024, 1024]
    assert res["loss_last"] < 0.3 * res["loss_first"], res
    assert res["psnr_test"] > res["psnr_untrained"] + 6.0, res


def test_an_overflowed_iteration_changes_neither_the_net_nor_the_pose_rows():
    """An iteration whose forward pass overflowed its pair buffer yields zero rasterizer gradients and a raised device flag;
    `AvatarModel.step` must then leave EVERYTHING as it was: the net / geometry features (optim.Adam reads the flag) and the
    pose / translation rows of the batch with their SparseAdam moments (a SparseAdam step on the zero gradients
    would still decay the moments and apply a momentum-only update — `_pose_step` puts the rows back). The next, valid
    iteration moves all of them again."""
    import time
    from gaussianavatar_amd import rasterizer as R
    from gaussianavatar_amd.avatar_model import collate_frames
    from gaussianavatar_amd.losses import l1_loss_w
    m, mp, npar, op = _small_model()
    epoch = op.pose_op_start_iter + 1                      # pose optimisation active
    batch = collate_frames([m.train_dataset[i] for i in range(2)], "cuda")
    gt = torch.full((2, 3, 96, 96), 0.4, device="cuda")

    def iteration():
        image, *_ = m.train_stage1(batch, 40)
        loss = l1_loss_w(image, gt)
        m.zero_grad(epoch)
        loss.backward()
        m.step(epoch)

    def state():
        sp = m.optimizer_pose.state
        out = [p.detach().clone() for p in m.net.parameters()] + [m.geo_feature.detach().clone(),
                                                                  m.pose.weight.detach().clone(), m.transl.weight.detach().clone()]
        for p in (m.pose.weight, m.transl.weight):
            out += [sp[p][k].clone() for k in ("exp_avg", "exp_avg_sq") if p in sp and k in sp[p]]
        return out

    iteration(); iteration()                               # moments exist, capacity history exists
    R.check_overflow(block=True)
    key = (m.query_points.shape[1], 96, 96)
    saved = (R._capacity.pairs_per_gaussian, R._capacity.floor, dict(R._capacity.seen))
    try:
        R._capacity.pairs_per_gaussian, R._capacity.floor = 0, 64
        R._capacity.seen[key] = 10                         # stale history: the buffer is far too small, nobody checks
        R._capacity.stamp[key] = time.monotonic()
        before = state()
        with pytest.warns(UserWarning, match="pair buffer overflow"):
            iteration()
            R.check_overflow(block=True)
        after = state()
        assert len(before) == len(after)
        for a, b in zip(before, after):
            assert torch.equal(a, b)
        assert int(R.overflow_flag("cuda")) == 0
        iteration()                                        # the capacity has adapted: a valid step
        R.check_overflow(block=True)
        moved = state()
        assert not torch.equal(before[0], moved[0]) and not torch.equal(before[-3], moved[-3])
    finally:
        R._capacity.pairs_per_gaussian, R._capacity.floor = saved[0], saved[1]
        R._capacity.seen = saved[2]
        R._capacity.pending.clear()
        R.clear_overflow_flag()
