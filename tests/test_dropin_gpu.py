"""The reference's OWN caller code, byte for byte (tests/golden/reference_scripts/), executed on the GPU
against this repository:

  * gaussian_renderer/__init__.py (the 50-line shim over `diff_gaussian_rasterization`) — image and
    gradients against the C oracle;
  * train.py, stage 1 and stage 2 — through gaussianavatar_amd.run_reference (dropin/ import aliases); the
    first iteration's image AND the parameter gradients its `loss.backward()` produced are compared with an
    all-CPU evaluation of the reference's formulas (tests/cpu_reference.py), and the run must fit;
  * eval.py and render_novel_pose.py on the checkpoint train.py wrote.

Third-party packages the scripts import but the image lacks (lpips, open3d, torchvision, torchmetrics) come
from tests/stubs (appended to the END of sys.path). If any fixture needed an edit to pass, the drop-in claim
would be false.
"""
import glob
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "reference_scripts")
STUBS = os.path.join(ROOT, "tests", "stubs")


class _Env:
    """sys.path / sys.modules / stdout / cwd as a reference script would see them; undone on exit."""

    def __enter__(self):
        from gaussianavatar_amd import run_reference
        self.rr = run_reference
        self.saved = (list(sys.path), set(sys.modules), sys.stdout, os.getcwd(), sys.argv)
        run_reference.install_paths()
        if STUBS not in sys.path:
            sys.path.append(STUBS)
        return self

    def __exit__(self, *exc):
        path, mods, stdout, cwd, argv = self.saved
        sys.path[:] = path
        sys.stdout, sys.argv = stdout, argv
        os.chdir(cwd)
        drop = self.rr.ALIASES + ("lpips", "open3d", "torchvision", "torchmetrics")
        for name in set(sys.modules) - mods:
            if name.split(".")[0] in drop:
                del sys.modules[name]
        return False


def test_reference_gaussian_renderer_shim_verbatim_vs_oracle(raster_oracle):
    """exec() of the reference's gaussian_renderer/__init__.py; its `from diff_gaussian_rasterization import
    ...` binds to the HIP rasterizer. Called the way the reference's model calls it (0-d CUDA tensors for
    FoV / height / width, model/avatar_model.py:333-336)."""
    from tests.scenes import cam_kwargs, random_scene
    from tests.hip_helpers import scene_tensors, settings_from_scene
    path = os.path.join(FIX, "gaussian_renderer__init__.py.txt")
    with _Env():
        shim = types.ModuleType("gaussian_renderer")
        exec(compile(open(path).read(), path, "exec"), shim.__dict__)
        for kind, P in (("avatar", 3000), ("general", 1500)):
            sc = random_scene(P, 112, 80, seed=5, kind=kind)
            t = scene_tensors(sc, requires_grad=True)
            rs = settings_from_scene(sc)
            dev = lambda v: torch.tensor(v, device="cuda")
            img = shim.render_batch(t["means3D"], None, t["colors"], t["rotations"], t["scales"], t["opacities"],
                                    dev(2 * math.atan(sc["tanfovx"])), dev(2 * math.atan(sc["tanfovy"])),
                                    dev(80), dev(112), rs.bg, rs.viewmatrix, rs.projmatrix, 0, rs.campos)
            ref = raster_oracle.forward(sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"],
                                        **cam_kwargs(sc))
            assert img.shape == (3, 80, 112)
            assert float(np.abs(img.detach().cpu().numpy() - ref["color"]).mean()) <= 1e-4
            g = np.random.default_rng(1).normal(0, 1, ref["color"].shape).astype(np.float32)
            img.backward(torch.tensor(g, device="cuda"))
            rb = raster_oracle.backward(ref, g)
            for k, key in (("means3D", "dmeans3D"), ("colors", "dcolors"), ("scales", "dscales"), ("rotations", "drots")):
                err = np.abs(t[k].grad.cpu().numpy() - rb[key]).max() / (np.abs(rb[key]).max() + 1e-12)
                assert err <= 2e-3, (kind, k, err)
            err = np.abs(t["opacities"].grad.cpu().numpy().reshape(-1) - rb["dopacity"]).max() / np.abs(rb["dopacity"]).max()
            assert err <= 2e-3, (kind, "opacity", err)


# --------------------------------------------------------------------------------- train.py & co.
EPOCHS = 10
WIDTHS = ["--c_geom", "64", "--c_pose", "64", "--hsize", "128", "--nf", "32"]      # production widths


def _write_dataset(tmp, smpl_type="smpl"):
    from gaussianavatar_amd.synthetic import make_assets, make_frames, write_dataset
    assets = make_assets(3000, 64, smpl_type)
    frames = make_frames(assets, 4, 96, 96)
    images = torch.ones(4, 3, 96, 96)
    images[:, :, 24:76, 38:58] = torch.tensor([0.25, 0.4, 0.6]).view(1, 3, 1, 1)
    return write_dataset(os.path.join(tmp, "data"), os.path.join(tmp, "proj"), assets, frames, images=images,
                         inp_posmap_size=64, stage2=True)


def _run_train(tmp, paths, stage, out, extra=()):
    """Runs the train.py fixture; records the first iteration (state before it, batch, image, the gradients
    its backward left in .grad) and every L1 value the loop computed."""
    from tests import cpu_reference
    import gaussianavatar_amd.avatar_model as AM
    import gaussianavatar_amd.losses as L
    rec = {"l1": []}
    fn_name = "train_stage1" if stage == 1 else "train_stage2"
    orig_fn, orig_step, orig_l1 = getattr(AM.AvatarModel, fn_name), AM.AvatarModel.step, L.l1_loss_w

    def stage_fn(self, batch, it):
        first = "snap" not in rec
        if first:
            rec.update(snap=cpu_reference.snapshot(self), it=it, model=self,
                       batch={k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        out_ = orig_fn(self, batch, it)
        if first:
            rec["image"] = out_[0].detach().cpu().clone()
        return out_

    def step(self, epoch):
        if "grads" not in rec:
            mods = {"net": self.net}
            if stage == 2:
                mods["pose_encoder"] = self.pose_encoder
            rec["grads"] = {m + "." + k: p.grad.detach().cpu().clone() for m, mod in mods.items()
                            for k, p in mod.named_parameters() if p.grad is not None}
            rec["geo_grad"] = self.geo_feature.grad.detach().cpu().clone()
            rec["epoch"] = epoch
        return orig_step(self, epoch)

    def l1(a, b):
        v = orig_l1(a, b)
        rec["l1"].append(float(v))
        return v

    argv = ["-s", paths["source_path"], "-m", out, "--project_path", paths["project_path"],
            "--smpl_model_path", paths["smpl_model_path"], "--smplx_model_path", paths["smplx_model_path"],
            "--test_folder", paths["test_folder"], "--epochs", str(EPOCHS), "--batch_size", "2",
            "--query_posmap_size", "64", "--inp_posmap_size", "64", "--save_epoch", "5", "--save_epochs", "0",
            "--train_stage", str(stage), "--quiet"] + WIDTHS + list(extra)
    setattr(AM.AvatarModel, fn_name, stage_fn)
    AM.AvatarModel.step = step
    L.l1_loss_w = l1
    try:
        with _Env() as env:
            os.chdir(tmp)
            env.rr.run(os.path.join(FIX, "train.py.txt"), argv)
    finally:
        setattr(AM.AvatarModel, fn_name, orig_fn)
        AM.AvatarModel.step = orig_step
        L.l1_loss_w = orig_l1
    return rec


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("dropin"))
    paths = _write_dataset(tmp)
    out1 = os.path.join(tmp, "out_stage1")
    rec1 = _run_train(tmp, paths, 1, out1)
    return dict(tmp=tmp, paths=paths, out1=out1, rec1=rec1)


def _check_first_iteration(rec, stage, raster_oracle, lambda_dssim=0.2, lambda_scale=3e-2, lambda_rgl=10.0):
    """train.py:66-89 on the CPU with the reference's formulas -> image and gradients."""
    from tests import cpu_reference
    from gaussianavatar_amd.losses import l1_loss_w, ssim
    model, snap, batch = rec["model"], rec["snap"], rec["batch"]
    ref = cpu_reference.forward(model, snap, batch, rec["it"], raster_oracle, stage=stage)
    l1_img = float((rec["image"] - ref["image"].detach()).abs().mean())
    assert l1_img <= 1e-4, l1_img
    gt = batch["original_image"].float()
    img = ref["image"]
    loss = (1.0 - lambda_dssim) * l1_loss_w(img, gt) + lambda_dssim * (1.0 - ssim(img, gt)) + lambda_rgl * ref["offset_loss"]
    loss = loss + (lambda_scale * ref["scale_loss"] + ref["geo_loss"] if stage == 1 else 10 * ref["pose_loss"])
    loss.backward()
    cpu = {"net." + k: p.grad for k, p in snap["net"].named_parameters() if p.grad is not None}
    if stage == 2:
        cpu.update({"pose_encoder." + k: p.grad for k, p in snap["pose_encoder"].named_parameters() if p.grad is not None})
    assert set(cpu) == set(rec["grads"]), set(cpu) ^ set(rec["grads"])
    gmax = max(float(g.abs().max()) for g in cpu.values())
    worst = ("", 0.0)
    for k, g in cpu.items():
        err = float((rec["grads"][k] - g).abs().max())
        tol = 5e-3 * float(g.abs().max()) + 2e-4 * gmax
        worst = max(worst, (k, err / tol), key=lambda t: t[1])
        assert err <= tol, (k, err, tol)
    err = float((rec["geo_grad"] - snap["geo"].grad).abs().max())
    assert err <= 5e-3 * float(snap["geo"].grad.abs().max()), err
    return l1_img, worst


def test_reference_train_py_stage1_verbatim(trained, raster_oracle):
    rec = trained["rec1"]
    assert len(rec["l1"]) == 2 * EPOCHS                    # 4 frames / batch 2, every iteration ran
    assert np.isfinite(rec["l1"]).all()
    assert np.mean(rec["l1"][-4:]) < np.mean(rec["l1"][:4]), rec["l1"]
    assert rec["it"] == 1 and rec["epoch"] == 1
    _check_first_iteration(rec, 1, raster_oracle)
    out = trained["out1"]
    assert os.path.exists(os.path.join(out, "cfg_args"))
    for ep in (5, 10):
        assert os.path.exists(os.path.join(out, "net", "iteration_%d" % ep, "net.pth"))
    # first log step of the loop (train.py:103-113): point clouds + prediction / ground-truth images
    assert os.path.exists(os.path.join(out, "log", "pred_0.ply")) and os.path.exists(os.path.join(out, "log", "00001_pred.png"))


def test_reference_eval_and_novel_pose_scripts_verbatim(trained):
    tmp, out = trained["tmp"], trained["out1"]
    with _Env() as env:
        os.chdir(tmp)
        env.rr.run(os.path.join(FIX, "eval.py.txt"), ["-m", out, "--epoch", "10", "--quiet"])
    txt = open(os.path.join(tmp, "results.txt")).read()
    psnr = float(txt.split("PSNR:")[1].split()[0])
    assert math.isfinite(psnr) and psnr > 3.0, txt      # 20 iterations of fitting: plumbing, not quality
    assert len(glob.glob(os.path.join(out, "test_free", "ours_10", "*.png"))) == 4
    with _Env() as env:
        os.chdir(tmp)
        env.rr.run(os.path.join(FIX, "render_novel_pose.py.txt"), ["-m", out, "--epoch", "10", "--quiet"])
    assert len(glob.glob(os.path.join(out, "novel_pose", "ours_10", "*.png"))) == 4


def test_reference_train_py_stage2_verbatim(trained, raster_oracle):
    """Stage 2 (pose-encoder UNet + decoder per frame, BatchNorm statistics over the batch), started from
    the stage-1 checkpoint the way the reference does (`--stage1_out_path`, train.py:43-44)."""
    tmp = trained["tmp"]
    out2 = os.path.join(tmp, "out_stage2")
    ckpt = os.path.join(trained["out1"], "net", "iteration_10")
    rec = _run_train(tmp, trained["paths"], 2, out2, extra=["--stage1_out_path", ckpt])
    assert len(rec["l1"]) == 2 * EPOCHS and np.isfinite(rec["l1"]).all()
    _check_first_iteration(rec, 2, raster_oracle)
    assert os.path.exists(os.path.join(out2, "net", "iteration_10", "pose_encoder.pth"))
