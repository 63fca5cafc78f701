"""The drop-in surface on the CPU: the reference's import paths resolve to this repository, the
command-line groups equal the reference's, and the script fixtures are the reference's bytes.
(The scripts themselves are executed on the GPU: tests/test_dropin_gpu.py.)"""
import hashlib
import importlib
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "reference_scripts")
REF = "/root/reference"
FIXTURES = {"gaussian_renderer__init__.py.txt": "gaussian_renderer/__init__.py", "train.py.txt": "train.py",
            "eval.py.txt": "eval.py", "render_novel_pose.py.txt": "render_novel_pose.py"}


@pytest.fixture
def dropin_paths():
    from gaussianavatar_amd import run_reference
    saved_path, saved_mods = list(sys.path), set(sys.modules)
    run_reference.install_paths()
    yield run_reference
    sys.path[:] = saved_path
    for name in set(sys.modules) - saved_mods:
        if name.split(".")[0] in run_reference.ALIASES:
            del sys.modules[name]


def test_fixtures_are_the_reference_bytes():
    sums = dict(line.split()[::-1] for line in open(os.path.join(FIX, "SHA256SUMS")).read().splitlines())
    for fx, ref in FIXTURES.items():
        data = open(os.path.join(FIX, fx), "rb").read()
        assert hashlib.sha256(data).hexdigest() == sums[fx], fx
        if os.path.isdir(REF):
            assert data == open(os.path.join(REF, ref), "rb").read(), fx


def test_reference_import_paths_resolve_here(dropin_paths):
    """Every first-party import of train.py / eval.py / render_novel_pose.py / model/avatar_model.py."""
    import gaussianavatar_amd as G
    wanted = {
        "model.avatar_model": ["AvatarModel"],
        "model.network": ["POP_no_unet"],
        "model.modules": ["UnetNoCond5DS", "GeomConvLayers", "ShapeDecoder", "uv_to_grid"],
        "scene.dataset_mono": ["MonoDataset_train", "MonoDataset_test", "MonoDataset_novel_pose", "MonoDataset_novel_view"],
        "utils.loss_utils": ["l1_loss_w", "ssim", "l2_loss"],
        "utils.general_utils": ["safe_state", "to_cuda", "adjust_loss_weights", "worker_init_fn", "load_masks",
                                "getIdxMap_torch"],
        "utils.graphics_utils": ["geom_transform_points", "getWorld2View2", "getProjectionMatrix", "focal2fov", "fov2focal"],
        "utils.system_utils": ["mkdir_p", "searchForMaxIteration"],
        "utils.image_utils": ["psnr", "mse"],
        "gaussian_renderer": ["render_batch"],
        "arguments": ["ModelParams", "OptimizationParams", "NetworkParams", "get_combined_args", "smpl_cpose_param"],
        "diff_gaussian_rasterization": ["GaussianRasterizationSettings", "GaussianRasterizer"],
    }
    for mod, names in wanted.items():
        m = importlib.import_module(mod)
        assert os.path.abspath(m.__file__).startswith(ROOT), (mod, m.__file__)
        for n in names:
            assert hasattr(m, n), (mod, n)
    import model.avatar_model as A
    from gaussianavatar_amd.avatar_model import AvatarModel
    assert A.AvatarModel is AvatarModel
    # the same surface as the reference's class (model/avatar_model.py)
    for meth in ("training_setup", "zero_grad", "step", "train_stage1", "train_stage2", "render_free_stage1",
                 "render_free_stage2", "save", "load", "stage_load", "stage2_load", "getTrainDataloader",
                 "getTestDataset", "getNovelposeDataset", "getNovelviewDataset", "net_set"):
        assert callable(getattr(AvatarModel, meth)), meth


def test_uv_index_map_and_projection_aliases_match_golden(dropin_paths):
    import numpy as np
    from utils.general_utils import getIdxMap_torch
    from utils.graphics_utils import geom_transform_points
    d = np.load(os.path.join(ROOT, "tests", "golden", "dataset_golden.npz"))
    np.testing.assert_array_equal(getIdxMap_torch(torch.rand(3, 8, 8)).numpy(), d["idx_map_8"])
    c = np.load(os.path.join(ROOT, "tests", "golden", "camera_loss_golden.npz"))
    out = geom_transform_points(torch.tensor(c["proj_pts"]), torch.tensor(c["full_1024"]))
    np.testing.assert_allclose(out.numpy(), c["proj_out"], rtol=1e-5, atol=1e-6)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (build container only)")
def test_argument_groups_equal_the_reference(dropin_paths):
    """Flags, short flags, types and defaults of the three parameter groups, against the reference's
    arguments/__init__.py imported with a stand-in for its (unused here) pytorch3d import."""
    from argparse import ArgumentParser
    import arguments as mine
    t = types.ModuleType("pytorch3d.transforms")
    t.euler_angles_to_matrix = lambda a, conv: torch.eye(3)[None]
    t.matrix_to_axis_angle = lambda m: torch.zeros(1, 3)
    p3 = types.ModuleType("pytorch3d")
    p3.transforms = t
    sys.modules["pytorch3d"], sys.modules["pytorch3d.transforms"] = p3, t
    try:
        spec = importlib.util.spec_from_file_location("_ref_arguments", os.path.join(REF, "arguments", "__init__.py"))
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        del sys.modules["pytorch3d"], sys.modules["pytorch3d.transforms"]

    def table(mod, sentinel):
        parser = ArgumentParser()
        mod.ModelParams(parser, sentinel=sentinel)
        mod.NetworkParams(parser)
        mod.OptimizationParams(parser)
        return {a.dest: (tuple(a.option_strings), a.default, getattr(a.type, "__name__", None), type(a).__name__)
                for a in parser._actions if a.dest != "help"}
    for sentinel in (False, True):
        assert table(mine, sentinel) == table(ref, sentinel)
    assert torch.equal(mine.smpl_cpose_param, ref.smpl_cpose_param)
    assert torch.equal(mine.smplx_cpose_param, ref.smplx_cpose_param)
    cmd = ["-s", "/tmp/d", "-m", "/tmp/o", "--train_stage", "2", "--epochs", "7", "-w"]
    for mod in (mine, ref):
        parser = ArgumentParser()
        groups = (mod.ModelParams(parser), mod.NetworkParams(parser), mod.OptimizationParams(parser))
        args = parser.parse_args(cmd)
        got = [vars(g.extract(args)) for g in groups]
        if mod is mine:
            first = got
    assert first == got
