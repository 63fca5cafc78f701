"""The ASSEMBLED training iteration at the sizes bench.py times: AvatarModel.train_stage1 /
train_stage2 at 200,000 Gaussians, a 512^2 UV map, 1024^2 images, 2 frames (the headline and the stage-2 line); stage 1 in the
warm-up regime (iteration 300: ~20 mm Gaussians, ~5 M pairs per frame, hundreds of tile lists beyond 8192 keys inside
train_stage1); and BASELINE config 5's own size (stage 2, SMPL-X, 300,000 Gaussians on a 1024^2 UV map, 1920 x 1080, one
frame = one GPU's share) — image, regulariser values and every
parameter / geometry-feature / pose gradient against the all-CPU evaluation of the reference's formulas
(tests/cpu_reference.py: /root/reference/model/avatar_model.py:272-367 stage 1, :369-463 stage 2; loss of
/root/reference/train.py:68-86; CPU nets pinned by the net goldens, oracle LBS, C-oracle rasterizer with its analytic
backward). The kernels are tested one by one elsewhere; this is the configuration of BENCH_rNN.json itself."""
import json
import os

import numpy as np
import pytest
import torch

from tests.grad_check import assert_grads_close

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name, payload):
    """measured errors next to the bars, for profiles/ (gpurun_out/ is merged back from the GPU box)"""
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"assembled_parity_{name}.json"), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


def _errors(pairs):
    rows = {}
    for n, a, b in pairs:
        a = a.detach().double().cpu().reshape(-1)
        b = b.detach().double().cpu().reshape(-1)
        tmax = float(b.abs().max())
        rows[n] = {"max_err": float((a - b).abs().max()), "tensor_max": tmax,
                   "cosine": float(a @ b / (a.norm() * b.norm() + 1e-300)) if tmax > 0 else None}
    return rows


# name -> (stage, smpl_type, Gaussians, UV edge, W, H, frames, iteration, scale-head bias or None)
CONFIGS = {
    "stage1": (1, "smpl", 200_000, 512, 1024, 1024, 2, 7, None),          # BENCH_rNN.json's own configuration
    "stage2": (2, "smpl", 200_000, 512, 1024, 1024, 2, 7, -5.65),         # bench.py's secondary line (3.5 mm stand-in)
    # the warm-up regime of a from-scratch training (/root/reference/model/avatar_model.py:315-316 at iteration 300): the
    # scale head stands at sigmoid(-2.64) x 0.3 = 20 mm — what the net has learnt by then (profiles/r06_raster_ubench.json:
    # warmup_300); at its random initialisation it would be 15 cm and 125 M pairs per frame, beyond what the CPU side renders
    "stage1_warmup300": (1, "smpl", 200_000, 512, 1024, 1024, 2, 300, -2.64),
    "config5": (2, "smplx", 300_000, 1024, 1920, 1080, 1, 7, -5.65),      # BASELINE config 5, one GPU's share
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_assembled_iteration_at_the_benchmarked_size_matches_cpu_reference(name, raster_oracle, raster_oracle_f64, monkeypatch):
    from gaussianavatar_amd import rasterizer
    from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
    from gaussianavatar_amd.losses import l1_loss_w, ssim
    from tests import cpu_reference
    from tests.test_raster_gpu import IMG_MAX_TOL, IMG_ROUND_TOL
    torch.manual_seed(0)
    stage, smpl_type, N, uv, W, H, B, iteration, scale_bias = CONFIGS[name]
    mp, npar, op = default_params(batch_size=B, num_points=N, image_width=W, image_height=H, num_frames=16,
                                  train_stage=stage, smpl_type=smpl_type, query_posmap_size=uv)
    m = AvatarModel(mp, npar, op, train=True)
    m.training_setup()
    m.net.train()
    if scale_bias is not None:
        with torch.no_grad():                       # bench.py's stand-in for a trained scale head
            m.net.decoder.conv8N.weight.mul_(0.01)
            m.net.decoder.conv8N.bias.fill_(scale_bias)
    batch = collate_frames([m.train_dataset[i] for i in range(B)], "cuda")
    gt = torch.ones(B, 3, H, W, device="cuda")
    gt[:, :, H // 5: 4 * H // 5, 2 * W // 5: 3 * W // 5] = 0.6
    l = op.lambda_dssim
    # one un-recorded iteration first: the pair capacity of this scene is then known (no re-render inside the checked one)
    with torch.no_grad():
        (m.train_stage1 if stage == 1 else m.train_stage2)(batch, iteration)
    rasterizer.check_overflow(block=True)
    snap = cpu_reference.snapshot(m)

    def objective(out_image, terms, gt_):
        if stage == 1:       # /root/reference/train.py:68-76
            return (op.lambda_scale * terms["scale_loss"] + op.lambda_rgl * terms["offset_loss"]
                    + (1.0 - l) * l1_loss_w(out_image, gt_) + l * (1.0 - ssim(out_image, gt_)) + terms["geo_loss"])
        return (op.lambda_rgl * terms["offset_loss"] + (1.0 - l) * l1_loss_w(out_image, gt_)
                + l * (1.0 - ssim(out_image, gt_)) + 10.0 * terms["pose_loss"])

    # what the model hands the rasterizer in the checked iteration (the batched entry point, renderer.render_frames)
    raster_calls = []
    batch_entry = rasterizer.rasterize_gaussians_batch
    monkeypatch.setattr(rasterizer, "rasterize_gaussians_batch",
                        lambda *a: (raster_calls.append(tuple(t.detach() if torch.is_tensor(t) else t for t in a)), batch_entry(*a))[1])
    hooked = []
    if stage == 1:
        image, pts, offset_loss, geo_loss, scale_loss = m.train_stage1(batch, iteration)
        terms = dict(offset_loss=offset_loss, geo_loss=geo_loss, scale_loss=scale_loss)
    else:
        # (the gradient that enters the pose encoder is compared too, and feeds the encoder's float64 check below)
        handle = m.pose_encoder.register_forward_hook(lambda mod, inp, out: (out.retain_grad(), hooked.append(out))[0])
        image, pts, pose_loss, offset_loss = m.train_stage2(batch, iteration)
        handle.remove()
        terms = dict(offset_loss=offset_loss, pose_loss=pose_loss)
    loss = objective(image, terms, gt)
    m.zero_grad(1)
    loss.backward()
    rasterizer.check_overflow(block=True)

    cpu_batch = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
    ref = cpu_reference.forward(m, snap, cpu_batch, iteration, raster_oracle, stage=stage)
    ref_loss = objective(ref["image"], ref, gt.cpu())
    ref_loss.backward()

    img_l1 = float((image.detach().cpu() - ref["image"].detach()).abs().mean())
    img_max = float((image.detach().cpu() - ref["image"].detach()).abs().max())
    pts_err = float((pts.detach().cpu() - ref["full_pred"].detach()).abs().max())
    scalars = {k: (float(v.detach()), float(ref[k].detach())) for k, v in terms.items()}
    pairs = [("net." + k, p.grad, dict(snap["net"].named_parameters())[k].grad) for k, p in m.net.named_parameters()]
    if stage == 2:
        pairs += [("enc." + k, p.grad, dict(snap["pose_encoder"].named_parameters())[k].grad)
                  for k, p in m.pose_encoder.named_parameters()]
    pairs.append(("geo", m.geo_feature.grad, snap["geo"].grad))
    pairs.append(("pose", m.pose.weight.grad.to_dense(), snap["pose"].grad))
    pairs.append(("transl", m.transl.weight.grad.to_dense(), snap["transl"].grad))
    for n, g, c in pairs:
        assert (g is None) == (c is None), n
    pairs = [(n, g, c) for n, g, c in pairs if c is not None]
    # The image's MAXIMUM norm. Between the assembled image and the all-CPU one it cannot be bounded:
    # the two rasterizers see inputs that differ by the nets' float32 rounding (positions to 4e-7, below), and a pixel's
    # colour is not continuous in them — two overlapping Gaussians whose depths differ by less than that swap their
    # blending order, a radius lands on the other side of a ceil() (measured: one pixel of the stage-2 frame pair moves by
    # 3.7 / 255). What CAN be bounded is split in two: (a) the rasterizer's inputs agree between the two sides; (b) on the
    # inputs the HIP rasterizer really got in this iteration its image obeys the bars of tests/test_raster_gpu.py:
    # assert_forward_parity against the C oracle — no pixel off by more than two alpha < 1/255 decisions (2 / 255), and
    # pixels off by more than rounding at most twice the oracle's own float32-vs-float64 disagreement on those inputs.
    diff = (image.detach().cpu() - ref["image"].detach()).abs().numpy()
    off_img = int((diff > IMG_ROUND_TOL).any(1).sum())
    g_means, g_cols, g_opac, g_scales, g_rots, _rs = raster_calls[-1]
    per_frame = lambda t, b, inner: (t[b] if (t.dim() == inner + 1 and t.shape[0] == B) else (t[0] if t.dim() == inner + 1 else t)).cpu().numpy()
    raster = dict(max_err=0.0, off=0, toss=0, scales_rel=0.0, colours_abs=0.0)
    for b, (mean_c, col_c, opa_c, sca_c, rot_c, cam_b) in enumerate(ref["raster_inputs"]):
        mean_g, col_g, sca_g = per_frame(g_means, b, 2), per_frame(g_cols, b, 2), per_frame(g_scales, b, 2)
        raster["scales_rel"] = max(raster["scales_rel"], float((np.abs(sca_g - sca_c) / np.abs(sca_c).clip(1e-12)).max()))
        raster["colours_abs"] = max(raster["colours_abs"], float(np.abs(col_g - col_c).max()))
        args = (mean_g, col_g, per_frame(g_opac, b, 2).reshape(-1), sca_g, per_frame(g_rots, b, 2))
        img32 = raster_oracle.forward(*args, **cam_b)["color"]
        img64 = raster_oracle_f64.forward(*args, **cam_b)["color"]
        d = np.abs(image[b].detach().cpu().numpy() - img32)
        raster["max_err"] = max(raster["max_err"], float(d.max()))
        raster["off"] += int((d > IMG_ROUND_TOL).any(0).sum())
        raster["toss"] += int((np.abs(img32.astype(np.float64) - img64) > IMG_ROUND_TOL).any(0).sum())
    status = rasterizer.last_status()
    _record(name, {"config": f"stage {stage} ({smpl_type}), {N} Gaussians, {uv}^2 UV, {W} x {H}, {B} frames, iteration {iteration}"
                             + (f", scale head bias {scale_bias}" if scale_bias is not None else ""),
                   "pairs_per_frame_max": status[0] if status else None, "longest_tile_list": status[3] if status else None,
                   "image_mean_l1": img_l1, "image_max_vs_all_cpu (not bounded: see the test)": img_max,
                   "pixels_off_by_more_than_rounding_vs_all_cpu": off_img,
                   "rasterizer_on_its_own_inputs_vs_oracle": {**raster, "max_bar": IMG_MAX_TOL},
                   "points_max_err": pts_err,
                   "loss": (float(loss.detach()), float(ref_loss.detach())), "scalars": scalars,
                   "gradients": _errors(pairs)})
    assert img_l1 <= 1e-4, img_l1
    assert raster["scales_rel"] <= 1e-4 and raster["colours_abs"] <= 1e-5, raster
    assert raster["max_err"] <= IMG_MAX_TOL, raster
    assert raster["off"] <= 2 * raster["toss"] + 4, raster
    assert pts_err <= 5e-5, pts_err
    for k, (a, b) in scalars.items():
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-9, (k, a, b)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-4 * abs(float(ref_loss.detach()))
    # per-tensor bar + cosine (tests/grad_check.py). The rasterizer's gradients are fp32 sums of ~800 k pair
    # contributions in non-deterministic (atomic) order on one side and in list order on the other.
    enc_pairs = [t for t in pairs if t[0].startswith("enc.")]
    rest = [t for t in pairs if not t[0].startswith("enc.")]
    if stage == 2:
        rest.append(("dL/d(pose features)", hooked[-1].grad, ref["pose_featmap"].grad))
    assert_grads_close(rest, rel=5e-3, floor=2e-4, cos_tol=1e-4)
    if enc_pairs:
        # The pose encoder is piecewise linear (ReLU / LeakyReLU on maps down to 4 x 4): an activation within float32
        # rounding of zero takes the other branch on one side, and that ONE element's gradient differs by its full value.
        # Measured at this size (tools/unet_check3.py): of the 65,536 inputs of upconv2's ReLU one is 1.4e-7 — the
        # gradient behind that gate is 4.6 % of the tensor's maximum off while the gradient in front of it agrees to
        # 6e-7, and upconv1's weight gradient (fed through that gate) moves by 5 % of its maximum; every convolution,
        # transposed convolution and BatchNorm of the encoder agrees with float64 to 1e-6 on its own inputs. A maximum
        # norm cannot hold such a tensor; the relative L2 error can: <= 1 % (cosine >= 1 - 5e-5) per tensor, and the
        # tensors behind the last gate (upconv2..5) to the usual per-element bar.
        for n, g, c in enc_pairs:
            a, b = g.detach().cpu().double().reshape(-1), c.double().reshape(-1)
            assert float(a @ b / (a.norm() * b.norm() + 1e-300)) >= 1.0 - 5e-5, n
        smooth = [t for t in enc_pairs if t[0].startswith(("enc.upconv2", "enc.upconv3", "enc.upconv4", "enc.upconv5"))]
        assert_grads_close(smooth + [t for t in rest if t[0] == "geo"], rel=5e-3, floor=2e-4, cos_tol=1e-4)
