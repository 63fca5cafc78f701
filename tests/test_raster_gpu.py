"""GPU parity tests: HIP rasterizer (through the C ABI) vs the CPU oracle.

Bars (BASELINE.md §2 / north_star): integer outputs (radii, tile rects, tiles_touched, per-tile
depth order) bit-exact; image within 1e-4 mean per-pixel L1; gradients within fp32 summation
noise of the oracle's analytic backward."""
import numpy as np
import pytest

from tests.scenes import cam_kwargs, random_scene

pytestmark = pytest.mark.gpu

IMG_L1_TOL = 1e-4          # mean |diff| per pixel-channel (north_star tolerance)
IMG_MAX_TOL = 2.0 / 255    # a borderline alpha < 1/255 decision moves a pixel by at most T/255; two of them in one pixel
IMG_ROUND_TOL = 1e-5       # what pure rounding (different association of the same blends) may move a pixel by
GRAD_REL_TOL = 2e-3        # relative to the largest |gradient| of that tensor

_oracle64 = []


def oracle64():
    """The float64 build of the oracle: the yardstick for how many per-pixel decisions of a scene are float32 coin
    tosses (the bars of assert_forward_parity are derived from its disagreement with the float32 oracle)."""
    if not _oracle64:
        from oracle.gsr_oracle import RasterOracle
        _oracle64.append(RasterOracle(f64=True))
    return _oracle64[0]


def oracle_forward(oracle, sc):
    return oracle.forward(sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"],
                          **cam_kwargs(sc), scale_modifier=sc.get("scale_modifier", 1.0))


def assert_forward_parity(oracle, sc, max_pairs=None):
    from tests.hip_helpers import hip_forward_state, hip_tile_lists
    ref = oracle_forward(oracle, sc)
    got = hip_forward_state(sc, max_pairs=max_pairs)
    assert got["status"][1] == 0, "unexpected overflow"
    assert got["status"][0] == ref["D"]
    np.testing.assert_array_equal(got["radii"], ref["radii"])
    np.testing.assert_array_equal(got["rect"], ref["rect"])
    np.testing.assert_array_equal(got["tiles_touched"].astype(np.uint32), ref["tiles_touched"])
    # tile ranges and per-tile order
    np.testing.assert_array_equal(got["tile_offset"][:-1].astype(np.uint32), ref["ranges"][:, 0])
    np.testing.assert_array_equal(got["tile_offset"][1:].astype(np.uint32), ref["ranges"][:, 1])
    np.testing.assert_array_equal(got["point_list"][:ref["D"]].astype(np.uint32), ref["point_list"])
    vis = ref["radii"] > 0
    np.testing.assert_allclose(got["xy"][vis], ref["xy"][vis], rtol=0, atol=0)
    np.testing.assert_allclose(got["conic_opacity"][vis], ref["conic_opacity"][vis], rtol=0, atol=0)
    diff = np.abs(got["color"] - ref["color"])
    assert diff.mean() <= IMG_L1_TOL, diff.mean()
    assert diff.max() <= IMG_MAX_TOL, diff.max()
    # How many pixels may differ by more than rounding, and how many n_contrib values may differ at all, is not a
    # constant: it is the number of per-pixel decisions (alpha < 1/255, T < 1e-4) that are float32 coin tosses in THIS
    # scene — measured as the disagreement of the float32 oracle with the float64 one. The HIP kernels evaluate
    # transmittance as a prefix-product tree, i.e. a third rounding of the same numbers: allowed twice that count.
    ref64 = oracle_forward(oracle64(), sc)
    toss_img = int((np.abs(ref["color"].astype(np.float64) - ref64["color"]) > IMG_ROUND_TOL).any(0).sum())
    toss_n = int((ref["n_contrib"].astype(np.int64) != ref64["n_contrib"].astype(np.int64)).sum())
    off_img = int((diff > IMG_ROUND_TOL).any(0).sum())
    off_n = int((got["n_contrib"].astype(np.int64) != ref["n_contrib"].astype(np.int64)).sum())
    assert off_img <= 2 * toss_img + 4, (off_img, toss_img)
    assert off_n <= 2 * toss_n + 4, (off_n, toss_n)
    return ref, got


@pytest.mark.parametrize("kind", ["general", "avatar"])
@pytest.mark.parametrize("P,W,H,scale", [(300, 48, 32, 0.05), (2000, 128, 96, 0.03), (5000, 256, 256, 0.02),
                                          (4000, 200, 120, 0.02)])
def test_forward_parity(raster_oracle, kind, P, W, H, scale):
    sc = random_scene(P, W, H, seed=P + W, kind=kind, scale_med=scale)
    assert_forward_parity(raster_oracle, sc)


@pytest.mark.parametrize("kind", ["general", "avatar"])
@pytest.mark.parametrize("P,W,H,scale", [(300, 48, 32, 0.05), (3000, 128, 128, 0.03), (4000, 200, 120, 0.02)])
def test_backward_parity(raster_oracle, raster_oracle_f64, kind, P, W, H, scale):
    """Every gradient tensor per ELEMENT against the float64 oracle (tests/test_raster_hardening_gpu.py:
    assert_gradient_elements; VERDICT r04 weak 1c: the tensor-max bar that used to live here let entries far below the
    maximum be arbitrarily wrong), bit-exact radii, exact zeros for invisible Gaussians."""
    from tests.hip_helpers import hip_forward_backward
    from tests.test_raster_hardening_gpu import assert_gradient_elements
    sc = random_scene(P, W, H, seed=7 + P, kind=kind, scale_med=scale)
    g = np.random.default_rng(3).normal(0, 1, (3, H, W)).astype(np.float32)
    ref = oracle_forward(raster_oracle, sc)
    rb = raster_oracle.backward(ref, g)
    rb64 = raster_oracle_f64.backward(oracle_forward(raster_oracle_f64, sc), g)
    color, radii, grads = hip_forward_backward(sc, g)
    np.testing.assert_array_equal(radii, ref["radii"])
    invisible = ref["radii"] == 0
    # (an avatar-like scene has unit opacity and identity rotations: their gradients are dead outputs of that path)
    checked = ("dmeans3D", "dcolors", "dopacity", "dscales", "drots") if kind == "general" else ("dmeans3D", "dcolors", "dscales")
    for k in ("dmeans3D", "dmeans2D", "dcolors", "dopacity", "dscales", "drots"):
        a, b = grads[k], rb[k]
        if k in checked:
            assert_gradient_elements(k, a, b, rb64[k])
        elif k == "dmeans2D":
            assert_gradient_elements(k, a[:, :2], b[:, :2], rb64[k][:, :2])
        else:
            assert np.abs(a - b).max() <= GRAD_REL_TOL * (np.abs(b).max() + 1e-12), k
        assert np.all(a[invisible] == 0), k


def test_empty_and_culled(raster_oracle):
    from tests.hip_helpers import hip_forward_state
    sc = random_scene(64, 64, 48, seed=1)
    # everything behind the camera (camera centre is at z = 2.5 looking down -z)
    sc["means3D"][:, 2] += 10.0
    ref, got = assert_forward_parity(raster_oracle, sc)
    assert (got["radii"] == 0).all() and got["status"][0] == 0
    np.testing.assert_allclose(got["color"], 1.0)       # pure background
    sc0 = random_scene(0, 64, 48, seed=1)
    got0 = hip_forward_state(sc0)
    np.testing.assert_allclose(got0["color"], 1.0)


def test_depth_ties_broken_by_index(raster_oracle):
    sc = random_scene(600, 64, 64, seed=5, scale_med=0.04)
    sc["means3D"][200:400] = sc["means3D"][0:200]      # exact duplicates: identical depth bits
    assert_forward_parity(raster_oracle, sc)


def test_crowded_tile_uses_merge_path(raster_oracle):
    """More than 8192 entries in one tile: chunk sort in LDS + merge passes through HBM."""
    sc = random_scene(20000, 64, 64, seed=11, kind="general", spread=0.02, scale_med=0.004)
    ref, got = assert_forward_parity(raster_oracle, sc)
    assert got["status"][3] > 8192, got["status"]


def test_large_gaussians_cover_all_tiles(raster_oracle):
    sc = random_scene(200, 160, 112, seed=2, kind="general", scale_med=0.5)
    assert_forward_parity(raster_oracle, sc)


def test_a_few_screen_filling_gaussians_among_many_small_ones(raster_oracle):
    """Rectangles above GSR_BIG_RECT tiles are binned by their whole wave with direct atomics, the rest through the
    workgroup's LDS table — both in the same workgroups here; the per-tile lists must come out identical."""
    sc = random_scene(30000, 512, 512, seed=21, kind="avatar", scale_med=0.004)
    sc["scales"][::3000] = 0.8          # ten Gaussians that cover the whole image (1024 tiles each)
    sc["scales"][1500::3000] = 0.05     # and ten of ~100 tiles
    ref, got = assert_forward_parity(raster_oracle, sc)
    assert got["status"][0] > 10 * 1024


def test_overflow_is_flagged_and_recovered(raster_oracle):
    import torch
    from gaussianavatar_amd import rasterizer as R
    from tests.hip_helpers import hip_forward_state, scene_tensors, settings_from_scene
    sc = random_scene(3000, 128, 128, seed=4, scale_med=0.03)
    ref = oracle_forward(raster_oracle, sc)
    got = hip_forward_state(sc, max_pairs=ref["D"] // 2)
    assert got["status"][1] == 1 and got["status"][0] == ref["D"]
    # evaluation-mode call re-renders transparently with a larger buffer
    saved = (R._capacity.pairs_per_gaussian, R._capacity.floor, dict(R._capacity.seen))
    key = (3000, 128, 128)
    try:
        R._capacity.pairs_per_gaussian, R._capacity.floor = 0, 64
        R._capacity.seen.pop(key, None)
        rs = settings_from_scene(sc)
        t = scene_tensors(sc)
        with torch.no_grad():
            color, radii = R.GaussianRasterizer(rs)(
                means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
                scales=t["scales"], rotations=t["rotations"])
        assert np.abs(color.cpu().numpy() - ref["color"]).mean() <= IMG_L1_TOL
        # the history holds the pair count, or the pair capacity the recorded segments asked for if larger
        assert R._capacity.seen[key] >= ref["D"] and R.last_status()[0] == ref["D"]
        # training mode, first call for an unknown shape: also exact (one-time synchronous check)
        R._capacity.seen.pop(key, None)
        t = scene_tensors(sc, requires_grad=True)
        color, radii = R.GaussianRasterizer(rs)(
            means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
            scales=t["scales"], rotations=t["rotations"])
        assert np.abs(color.detach().cpu().numpy() - ref["color"]).mean() <= IMG_L1_TOL
        # steady state with a stale history (scene suddenly needs > 2x the pairs): detected on a later poll
        R._capacity.seen[key] = 10
        R._capacity.stamp[key] = __import__("time").monotonic()
        color, radii = R.GaussianRasterizer(rs)(
            means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
            scales=t["scales"], rotations=t["rotations"])
        n_events = R.overflow_events()
        with pytest.warns(UserWarning, match="pair buffer overflow"):      # default policy: warn, keep running
            R.check_overflow(block=True)
        assert R.overflow_events() == n_events + 1
        # policy "raise" for callers that prefer to stop
        R._capacity.seen[key] = 10
        R._capacity.stamp[key] = __import__("time").monotonic()
        R.set_pair_capacity(on_overflow="raise")
        try:
            R.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                     colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
            with pytest.raises(R.RasterizerOverflow):
                R.check_overflow(block=True)
        finally:
            R.set_pair_capacity(on_overflow="warn")
        # ... after which the capacity has adapted
        color, radii = R.GaussianRasterizer(rs)(
            means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
            scales=t["scales"], rotations=t["rotations"])
        R.check_overflow(block=True)
        assert np.abs(color.detach().cpu().numpy() - ref["color"]).mean() <= IMG_L1_TOL
    finally:
        R._capacity.pairs_per_gaussian, R._capacity.floor = saved[0], saved[1]
        R._capacity.seen = saved[2]
        R._capacity.pending.clear()


def test_steady_state_overflow_never_reaches_the_parameters(raster_oracle):
    """A steady-state forward pass that overflows its pair buffer is detected by the host only iterations later (the
    status words are polled, never waited for). Until then nothing computed from the truncated tile lists may reach
    the parameters: the backward pass writes zeros and raises the device-side flag, optim.Adam's kernel reads it and
    changes nothing — parameters and moments bit-identical across that step — and training continues on the next
    one with the resized buffer (the reference resizes and never truncates)."""
    import time
    import torch
    from gaussianavatar_amd import rasterizer as R
    from gaussianavatar_amd.optim import Adam
    from tests.hip_helpers import scene_tensors, settings_from_scene
    sc = random_scene(3000, 128, 128, seed=4, kind="avatar", scale_med=0.03)
    ref = oracle_forward(raster_oracle, sc)
    key = (3000, 128, 128)
    saved = (R._capacity.pairs_per_gaussian, R._capacity.floor, dict(R._capacity.seen))
    try:
        R._capacity.pairs_per_gaussian, R._capacity.floor = 0, 64
        rs = settings_from_scene(sc)
        t = scene_tensors(sc, requires_grad=True)
        names = ("means3D", "colors", "scales")
        params = [t[k] for k in names]
        opt = Adam(params, lr=1e-3)

        def iteration():
            opt.zero_grad()
            color, _ = R.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                                colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
            ((color - 0.25) ** 2).mean().backward()
            return color.detach()

        R._capacity.seen.pop(key, None)
        iteration()                                   # first call of the shape: sized synchronously, no overflow
        opt.step()
        assert int(R.overflow_flag("cuda")) == 0
        state = lambda: [p.detach().clone() for p in params] + [opt.state[p][k].clone() for p in params
                                                                for k in ("exp_avg", "exp_avg_sq")]
        # steady state with a stale history: the scene needs far more pairs than the buffer holds
        R._capacity.seen[key] = 10
        R._capacity.stamp[key] = time.monotonic()
        before = state()
        with pytest.warns(UserWarning, match="pair buffer overflow"):     # (raised by whichever poll sees it first)
            iteration()
            assert int(R.overflow_flag("cuda")) == 1
            for p in params:
                assert float(p.grad.abs().max()) == 0.0     # no gradient from truncated lists
            opt.step()
            for a, b in zip(before, state()):
                assert torch.equal(a, b)                    # the step changed nothing
            # ... and lowered the flag behind itself: it describes ONE step, whatever clears the gradients
            assert int(R.overflow_flag("cuda")) == 0
            R.check_overflow(block=True)                    # the host learns of it at the latest here: capacity raised
        # training continues: the next step renders the full lists and moves the parameters
        color = iteration()
        assert int(R.overflow_flag("cuda")) == 0
        assert np.abs(color.cpu().numpy() - oracle_forward(
            raster_oracle, {**sc, **{k: t[k].detach().cpu().numpy() for k in names}})["color"]).mean() <= IMG_L1_TOL
        opt.step()
        R.check_overflow(block=True)
        assert not torch.equal(before[0], params[0].detach())
        assert ref["D"] > 64
    finally:
        R._capacity.pairs_per_gaussian, R._capacity.floor = saved[0], saved[1]
        R._capacity.seen = saved[2]
        R._capacity.pending.clear()
        R.clear_overflow_flag()


def test_forward_only_render_records_nothing_and_matches(raster_oracle):
    """torch.no_grad() / no input requires grad (the reference's eval.py:42,65 and render_novel_pose.py:30-32): the
    forward-only kernels — no segment records, the small workspace — must give the same image, bit for bit, as the
    recording render; per-frame and batched."""
    import torch
    from gaussianavatar_amd import _native, rasterizer as R
    from tests.hip_helpers import scene_tensors, settings_from_scene
    sc = random_scene(20000, 320, 200, seed=9, kind="general", scale_med=0.01)
    rs = settings_from_scene(sc)
    kw = lambda t: dict(means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
                        scales=t["scales"], rotations=t["rotations"])
    t = scene_tensors(sc, requires_grad=True)
    color_rec, radii_rec = R.GaussianRasterizer(rs)(**kw(t))
    with torch.no_grad():
        color_eval, radii_eval = R.GaussianRasterizer(rs)(**kw(t))
    t2 = scene_tensors(sc)                                  # nothing requires grad
    color_eval2, _ = R.GaussianRasterizer(rs)(**kw(t2))
    assert torch.equal(color_rec.detach(), color_eval) and torch.equal(color_rec.detach(), color_eval2)
    assert torch.equal(radii_rec, radii_eval)
    ref = oracle_forward(raster_oracle, sc)
    assert np.abs(color_eval.cpu().numpy() - ref["color"]).mean() <= IMG_L1_TOL
    # batched
    B = 3
    means = torch.stack([t2["means3D"] + 0.01 * f for f in range(B)])
    args = (t2["colors"], t2["opacities"], t2["scales"], t2["rotations"], rs)
    img_rec, _ = R.rasterize_gaussians_batch(means.clone().requires_grad_(True), *args)
    img_eval, _ = R.rasterize_gaussians_batch(means, *args)
    assert torch.equal(img_rec.detach(), img_eval)
    # the forward-only workspace is the small one
    lib = _native.gsr()
    cap = 1 << 20
    assert lib.gsr_workspace_bytes_for(20000, 320, 200, cap, _native.GSR_WS_EVAL) * 4 < \
        lib.gsr_workspace_bytes_for(20000, 320, 200, cap, _native.GSR_WS_TRAIN)


def test_api_errors_and_retain_graph():
    import torch
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    from tests.hip_helpers import scene_tensors, settings_from_scene
    sc = random_scene(500, 64, 64, seed=9)
    rs = settings_from_scene(sc)
    t = scene_tensors(sc, requires_grad=True)
    r = GaussianRasterizer(rs)
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(means3D=t["means3D"], means2D=None, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed"):
        r(means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"])
    color, _ = r(means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
                 scales=t["scales"], rotations=t["rotations"])
    loss = color.square().mean()
    loss.backward(retain_graph=True)
    g1 = t["means3D"].grad.clone()
    t["means3D"].grad = None
    loss.backward()
    g2 = t["means3D"].grad
    assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-7)
    vis = r.markVisible(t["means3D"].detach())
    assert vis.dtype == torch.bool and vis.shape == (500,)


def test_cov3d_precomp_path(raster_oracle):
    import torch
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    from tests.hip_helpers import scene_tensors, settings_from_scene
    sc = random_scene(800, 96, 64, seed=21, scale_med=0.04)
    ref = oracle_forward(raster_oracle, sc)
    cov = torch.tensor(ref["cov3d"], device="cuda", requires_grad=True)
    t = scene_tensors(sc, requires_grad=True)
    color, radii = GaussianRasterizer(settings_from_scene(sc))(
        means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
        cov3D_precomp=cov)
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    g = np.random.default_rng(0).normal(0, 1, ref["color"].shape).astype(np.float32)
    color.backward(torch.tensor(g, device="cuda"))
    rb = raster_oracle.backward(ref, g)
    err = np.abs(cov.grad.cpu().numpy() - rb["dcov3D"]).max() / (np.abs(rb["dcov3D"]).max() + 1e-12)
    assert err <= GRAD_REL_TOL, err


def test_batched_launch_equals_per_frame_calls(raster_oracle):
    """gsr_forward_batch / gsr_backward_batch (one launch for all frames, shared colours/scales read
    once) against the per-frame entry points."""
    import math
    import torch
    from gaussianavatar_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,
                                               rasterize_gaussians_batch)
    from tests.hip_helpers import settings_from_scene
    B, P, W, H = 3, 2500, 160, 112
    scs = [random_scene(P, W, H, seed=40 + b, kind="avatar", scale_med=0.03) for b in range(B)]
    rs0 = settings_from_scene(scs[0])
    dev = "cuda"
    means = torch.tensor(np.stack([s["means3D"] for s in scs]), device=dev, requires_grad=True)
    colors = torch.tensor(scs[0]["colors"], device=dev, requires_grad=True)          # shared
    scales = torch.tensor(scs[0]["scales"], device=dev, requires_grad=True)          # shared
    rots = torch.tensor(scs[0]["rotations"], device=dev)
    opac = torch.ones(P, 1, device=dev)
    views = rs0.viewmatrix[None].repeat(B, 1, 1).contiguous()
    views[1, 3, 0] += 0.05                                                            # per-frame cameras
    projs = torch.stack([v @ torch.linalg.inv(rs0.viewmatrix) @ rs0.projmatrix for v in views])
    g = torch.randn(B, 3, H, W, device=dev)
    rs = rs0._replace(viewmatrix=views, projmatrix=projs)
    img, radii = rasterize_gaussians_batch(means, colors[None].expand(B, -1, -1), opac,
                                           scales[None].expand(B, -1, -1), rots, rs)
    img.backward(g)
    got = (means.grad.clone(), colors.grad.clone(), scales.grad.clone())
    means.grad = colors.grad = scales.grad = None
    imgs = []
    for b in range(B):
        rsb = rs0._replace(viewmatrix=views[b], projmatrix=projs[b])
        im, rd = GaussianRasterizer(rsb)(means3D=means[b], means2D=None, opacities=opac, colors_precomp=colors,
                                         scales=scales, rotations=rots)
        assert torch.equal(rd, radii[b])
        imgs.append(im)
    ref = torch.stack(imgs)
    assert torch.equal(ref, img)
    ref.backward(g)
    for a, b_ in zip(got, (means.grad, colors.grad, scales.grad)):
        assert float((a - b_).abs().max()) <= 2e-4 * float(b_.abs().max()) + 1e-7


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_path(raster_oracle, deg):
    """`shs` instead of `colors_precomp` (rasterizer API completeness; the reference's avatar path
    always passes colors_precomp): colours, clamp flags and the gradients w.r.t. the SH
    coefficients and — through the view direction — the means, against the oracle."""
    import torch
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    from tests.hip_helpers import scene_tensors, settings_from_scene
    P, W, H, M = 1500, 96, 80, 16
    sc = random_scene(P, W, H, seed=40 + deg, scale_med=0.04)
    shs = np.random.default_rng(deg).normal(0, 1.2, (P, M, 3)).astype(np.float32)
    ref = raster_oracle.forward(sc["means3D"], None, sc["opacities"], sc["scales"], sc["rotations"],
                                **cam_kwargs(sc), shs=shs, sh_degree=deg, campos=sc["campos"])
    assert 0.02 < ref["sh"]["clamped"].mean() < 0.7
    rs = settings_from_scene(sc)._replace(sh_degree=deg)
    t = scene_tensors(sc, requires_grad=True)
    sht = torch.tensor(shs, device="cuda", requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                          shs=sht, scales=t["scales"], rotations=t["rotations"])
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    diff = np.abs(color.detach().cpu().numpy() - ref["color"])
    assert diff.mean() <= IMG_L1_TOL and diff.max() <= IMG_MAX_TOL, (diff.mean(), diff.max())
    g = np.random.default_rng(1).normal(0, 1, ref["color"].shape).astype(np.float32)
    color.backward(torch.tensor(g, device="cuda"))
    rb = raster_oracle.backward(ref, g)
    for got, want, name in ((sht.grad, rb["dsh"], "dsh"), (t["means3D"].grad, rb["dmeans3D"], "dmeans3D"),
                            (t["scales"].grad, rb["dscales"], "dscales")):
        err = np.abs(got.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-12)
        assert err <= GRAD_REL_TOL, (name, err)
    K = (deg + 1) ** 2
    assert torch.all(sht.grad[:, K:] == 0)           # coefficients beyond the active degree
    # view-direction term really is exercised: without it the means gradient differs
    if deg > 0:
        no_dir = rb["dmeans3D"] - raster_oracle.sh_backward(sc["means3D"], shs, deg, sc["campos"],
                                                            ref["sh"]["clamped"], rb["dcolors"])[1]
        assert np.abs(no_dir - rb["dmeans3D"]).max() > 1e-3 * np.abs(rb["dmeans3D"]).max()


def test_full_size_parity_and_invariants(raster_oracle):
    """BASELINE.json's headline size (200k avatar-like Gaussians, 1024 x 1024): direct parity with the
    oracle (it takes ~1 s per direction), plus size-independent invariants of the splatting algebra."""
    import torch
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    from tests.hip_helpers import hip_forward_backward, hip_tile_lists, scene_tensors, settings_from_scene
    P, W, H = 200_000, 1024, 1024
    sc = random_scene(P, W, H, seed=1, kind="avatar", spread=0.45, scale_med=0.0035)
    ref, got = assert_forward_parity(raster_oracle, sc)
    assert ref["D"] > 400_000
    # sortedness: every tile list ordered by (depth bits, index), keys unique
    depth_bits = ref["depth"].view(np.uint32).astype(np.uint64)
    for lst in hip_tile_lists(got)[::7]:
        if len(lst) > 1:
            key = (depth_bits[lst] << np.uint64(32)) | lst.astype(np.uint64)
            assert np.all(key[1:] > key[:-1])
    # checksum of checksums: pairs per tile add up to the Gaussians' tile counts
    assert int(got["tile_offset"][-1]) == int(ref["tiles_touched"].astype(np.int64).sum())
    # backward parity at full size
    g = np.random.default_rng(5).normal(0, 1, (3, H, W)).astype(np.float32)
    rb = raster_oracle.backward(ref, g)
    color, radii, grads = hip_forward_backward(sc, g)
    for k in ("dmeans3D", "dcolors", "dscales"):
        err = np.abs(grads[k] - rb[k]).max() / (np.abs(rb[k]).max() + 1e-12)
        assert err <= GRAD_REL_TOL, (k, err)
    # partition of unity: constant colour == background  =>  constant image (sum of weights + T = 1)
    sc1 = dict(sc, colors=np.full((P, 3), 0.37, np.float32), bg=np.full(3, 0.37, np.float32))
    t = scene_tensors(sc1, requires_grad=True)
    img, _ = GaussianRasterizer(settings_from_scene(sc1))(
        means3D=t["means3D"], means2D=None, opacities=t["opacities"], colors_precomp=t["colors"],
        scales=t["scales"], rotations=t["rotations"])
    assert float((img - 0.37).abs().max()) <= 2e-6
    # linear in the colours: d(sum image)/d colour_i summed over Gaussians = sum over pixels of (1 - T)
    img.sum().backward()
    lhs = t["colors"].grad.double().sum(0).cpu().numpy()
    rhs = float((1.0 - ref["final_T"].astype(np.float64)).sum())
    assert np.all(np.abs(lhs - rhs) <= 1e-4 * rhs), (lhs, rhs)
    # idempotence: a second forward reproduces the first bit for bit (no atomics in the forward image)
    img2, _ = GaussianRasterizer(settings_from_scene(sc1))(
        means3D=t["means3D"].detach(), means2D=None, opacities=t["opacities"].detach(),
        colors_precomp=t["colors"].detach(), scales=t["scales"].detach(), rotations=t["rotations"].detach())
    assert torch.equal(img.detach(), img2)


def test_1080p_parity_with_partial_tile_row(raster_oracle):
    """BASELINE.json's stage-2 image size: 1920 x 1080 is 120 x 68 tiles whose last row is half
    outside the image (1080 = 67.5 * 16); 300k Gaussians of the general kind (anisotropic, rotated,
    translucent) spread over the whole frame, forward and backward against the oracle."""
    from tests.hip_helpers import hip_forward_backward
    P, W, H = 300_000, 1920, 1080
    sc = random_scene(P, W, H, seed=21, kind="general", spread=1.2, scale_med=0.006)
    ref, got = assert_forward_parity(raster_oracle, sc)
    assert got["tile_offset"].shape[0] == 120 * 68 + 1
    last_row = got["tile_offset"][120 * 67:]
    assert int(last_row[-1] - last_row[0]) > 0          # the partial tile row is populated
    g = np.random.default_rng(8).normal(0, 1, (3, H, W)).astype(np.float32)
    rb = raster_oracle.backward(ref, g)
    _c, _r, grads = hip_forward_backward(sc, g)
    for k in ("dmeans3D", "dcolors", "dscales", "dopacity", "drots"):
        err = np.abs(grads[k] - rb[k]).max() / (np.abs(rb[k]).max() + 1e-12)
        assert err <= GRAD_REL_TOL, (k, err)


def test_more_than_8192_tiles_unordered_tile_walk(raster_oracle):
    """Above 8192 tiles K2 leaves the tile order unsorted: the sort / merge grids must not rely on it."""
    sc = random_scene(2000, 2064, 2064, seed=21, kind="general", scale_med=0.004)
    assert ((2064 + 15) // 16) ** 2 > 8192
    assert_forward_parity(raster_oracle, sc)
