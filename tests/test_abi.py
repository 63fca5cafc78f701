"""CPU tests: the C-ABI libraries load and export every symbol the headers declare; layout
queries work without a GPU; the drop-in package exposes the reference's names."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:gsr|galbs|ganet)_[a-z_0-9]+)\s*\(", src)))


@pytest.mark.parametrize("header,loader", [("gsr.h", "gsr"), ("galbs.h", "galbs"), ("ganet.h", "ganet")])
def test_every_declared_symbol_is_exported(header, loader):
    from gaussianavatar_amd import _native
    lib = getattr(_native, loader)()
    names = _declared(header)
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), n
    listed = {"gsr": _native.GSR_SYMBOLS, "galbs": _native.GALBS_SYMBOLS, "ganet": _native.GANET_SYMBOLS}[loader]
    assert set(names) == set(listed), set(names) ^ set(listed)


def test_layout_is_consistent_without_gpu():
    from gaussianavatar_amd import _native
    lib = _native.gsr()
    L = _native.GsrLayout()
    assert lib.gsr_workspace_layout(1000, 100, 70, 5000, ctypes.byref(L)) == 0
    assert L.total_bytes == lib.gsr_workspace_bytes(1000, 100, 70, 5000)
    offs = [getattr(L, f) for f in _native._LAYOUT_FIELDS[1:] if not f.endswith("_bytes")]
    assert len(set(offs)) == len(offs) and all(o % 256 == 0 and o < L.total_bytes for o in offs)
    # the sub-arrays are ordered by who needs them: forward-only < + backward state < + deterministic backward
    assert 0 < L.eval_bytes < L.train_bytes < L.total_bytes
    for mode, want in ((_native.GSR_WS_EVAL, L.eval_bytes), (_native.GSR_WS_TRAIN, L.train_bytes),
                       (_native.GSR_WS_DEBUG, L.total_bytes)):
        assert lib.gsr_workspace_bytes_for(1000, 100, 70, 5000, mode) == want
    assert lib.gsr_workspace_bytes_for(1000, 100, 70, 5000, 3) == 0
    for f in ("depth", "point_list", "final_T", "n_contrib", "status", "xyext", "seg_count"):
        assert getattr(L, f) < L.eval_bytes, f
    for f in ("grad_acc", "seg_entries", "seg_ckpt", "seg_info", "pix_accum", "seg_list"):
        assert L.eval_bytes <= getattr(L, f) < L.train_bytes, f
    assert L.pair_grad == L.train_bytes
    # a forward-only workspace is several times smaller per (tile, Gaussian) pair of capacity
    big = lib.gsr_workspace_bytes_for(1000, 100, 70, 1 << 20, _native.GSR_WS_TRAIN)
    small = lib.gsr_workspace_bytes_for(1000, 100, 70, 1 << 20, _native.GSR_WS_EVAL)
    assert small < big // 4
    assert lib.gsr_render_block_edge() == 4
    assert lib.gsr_workspace_bytes(-1, 100, 70, 5000) == 0
    assert lib.gsr_workspace_layout(10, 0, 70, 5000, ctypes.byref(L)) != 0
    assert b"invalid" in lib.gsr_last_error()
    assert _native.galbs().galbs_joint_saved_floats(24) == 24 * 21


def test_forward_rejects_bad_arguments_before_touching_the_device():
    from gaussianavatar_amd import _native
    lib = _native.gsr()
    st = _native.GsrSettings(64, 64, 0.5, 0.5, 1.0, 0, 0, 0, None, None, None, None)
    for fwd in (lib.gsr_forward, lib.gsr_forward_eval):
        rc = fwd(ctypes.byref(st), 10, None, None, None, 0, None, None, None, None, None, 0, 100, None, None, None)
        assert rc == 1 and b"device pointers" in lib.gsr_last_error()
    # the profiler is a caller-owned object bound to the calling thread: no process-global state
    prof = ctypes.c_void_p(lib.gsr_profile_create())
    assert prof.value
    assert lib.gsr_profile_bind(prof, 0x7f) == 0 and lib.gsr_profile_bind(None, 0) == 0
    ms, n = (ctypes.c_double * 7)(), (ctypes.c_int64 * 7)()
    assert lib.gsr_profile_read(prof, ms, n, 1) == 0 and sum(n) == 0
    assert lib.gsr_profile_read(None, ms, n, 1) == 1
    lib.gsr_profile_destroy(prof)


def test_dropin_package_surface():
    import diff_gaussian_rasterization as d
    fields = d.GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                      "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    assert callable(d.GaussianRasterizer) and hasattr(d.GaussianRasterizer, "markVisible")


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gaussianavatar_amd")):
        for f in files:
            if f.endswith(".py") or f.endswith(".hip") or f.endswith(".h"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
    assert "oracle" not in open(os.path.join(ROOT, "diff_gaussian_rasterization", "__init__.py")).read()


def test_ops_fail_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gaussianavatar_amd.lbs import skin
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        skin(torch.zeros(4, 3), None, torch.zeros(4, 24), torch.zeros(1, 24, 4, 4))
    from gaussianavatar_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.ones(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(rs)(means3D=torch.zeros(4, 3), means2D=None, opacities=torch.ones(4, 1),
                               colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3),
                               rotations=torch.zeros(4, 4))


def test_net_kernels_reject_bad_arguments_before_touching_the_device():
    """The decoder / convolution entry points validate their arguments on the host first: error code + text,
    nothing launched (works without a GPU)."""
    from gaussianavatar_amd import _native
    lib = _native.ganet()
    # convolutions: workspace / packed-weight queries and argument checks
    assert lib.ganet_conv5_packed_bytes(3) == 3 * 2 * 25 * (3 * 4 * 2 * 64) * 16
    assert lib.ganet_conv5_packed_bytes(0) == 0
    assert lib.ganet_conv5_wgrad_workspace(1, 128, 128) > 0
    assert lib.ganet_conv5_wgrad_workspace(1, 128, 100) == 0              # W must be a multiple of 64
    assert lib.ganet_conv5_pack(0, None, None, None) == 1 and b"ganet_conv5_pack" in lib.ganet_last_error()
    assert lib.ganet_conv5_apply(1, 128, 100, None, None, 0, 0, None, None) == 1
    assert b"W % 64" in lib.ganet_last_error()
    assert lib.ganet_conv5_wgrad(1, 128, 128, None, None, None, None, 0, None) == 1
    # decoder layers: shape checks
    assert lib.ganet_mlp_fwd(0, 128, 0, 128, None, 0, None, 0, None, None, None, None, None, 0, None, None, 0, None) == 1
    assert b"ganet_mlp_fwd" in lib.ganet_last_error()
    assert lib.ganet_wgrad_act_workspace(262144, 128, 128) == 256 * (128 * 128 + 128) * 4
    # no process-global arithmetic switch any more (round-2 review): the symbol must be gone
    assert not hasattr(lib, "ganet_set_mfma_mode")
