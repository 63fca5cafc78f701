"""CPU tests: the oracles against the golden vectors produced by the reference's own python
(oracle/make_golden.py) and against each other."""
import os

import numpy as np
import pytest
import torch

from tests.scenes import cam_kwargs, random_scene

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_lbs_oracle_matches_reference_lbs_golden():
    from oracle import lbs_oracle as O
    g = np.load(os.path.join(GOLD, "lbs_golden.npz"))
    for model in ("smpl", "smplx"):
        t = lambda k: torch.tensor(g[f"{model}_{k}"])
        J = O.rest_joints(t("betas"), t("v_template"), t("shapedirs"), t("J_regressor"))
        A = O.joint_transforms(t("pose"), t("transl"), J, torch.tensor(g[f"{model}_parents"]))
        np.testing.assert_allclose(A.numpy(), g[f"{model}_A"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(O.rodrigues(torch.tensor(g["smpl_pose"]).view(-1, 3)).numpy(),
                               g["smpl_rodrigues"], atol=1e-6)
    A = O.joint_transforms.__wrapped__ if hasattr(O.joint_transforms, "__wrapped__") else None
    assert A is None


def test_lbs_full_oracle_matches_reference_lbs_vertices_and_joints():
    """The whole of lbs() — pose blend shapes, per-vertex blend, vertex skinning — as bench.py's cpu_baseline times it
    ("as the reference runs it"): vertices, posed joints and A against the reference's own outputs."""
    from oracle import lbs_oracle as O
    g = np.load(os.path.join(GOLD, "lbs_golden.npz"))
    for model in ("smpl", "smplx"):
        t = lambda k: torch.tensor(g[f"{model}_{k}"])
        verts, joints, A = O.lbs_full(t("betas"), t("pose"), t("v_template"), t("shapedirs"), t("posedirs"),
                                      t("J_regressor"), torch.tensor(g[f"{model}_parents"]), t("lbs_weights"))
        np.testing.assert_allclose(verts.numpy(), g[f"{model}_verts"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(joints.numpy(), g[f"{model}_joints"], rtol=0, atol=5e-6)
        A = A.clone()
        A[:, :, :3, 3] += t("transl")[:, None]
        np.testing.assert_allclose(A.numpy(), g[f"{model}_A"], rtol=0, atol=2e-6)


def test_lbs_oracle_chain_matches_batch_rigid_transform_golden():
    """batch_rigid_transform with arbitrary rotation matrices (lbs.py:349-405)."""
    from oracle import lbs_oracle as O
    g = np.load(os.path.join(GOLD, "lbs_golden.npz"))
    R, J = torch.tensor(g["chain_R"]), torch.tensor(g["chain_J"])
    parents = torch.tensor(g["smpl_parents"])
    # feed the oracle rotations by inverting rodrigues is overkill: re-run its chain directly
    B, Jn = R.shape[:2]
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T = torch.zeros(B, Jn, 4, 4)
    T[:, :, :3, :3], T[:, :, :3, 3], T[:, :, 3, 3] = R, rel, 1
    chain = [T[:, 0]]
    for i in range(1, Jn):
        chain.append(chain[int(parents[i])] @ T[:, i])
    G = torch.stack(chain, 1)
    np.testing.assert_allclose(G[:, :, :3, 3].numpy(), g["chain_posed"], atol=2e-6)


def test_skin_oracle_matches_reference_einsums_golden():
    from oracle import lbs_oracle as O
    g = np.load(os.path.join(GOLD, "skin_golden.npz"))
    t = lambda k: torch.tensor(g[k])
    np.testing.assert_allclose(O.cano2live(t("A"), t("inv_mats")).numpy(), g["cano2live"], atol=1e-6)
    out = O.skin(t("query_points"), t("res"), t("weights"), t("cano2live"))
    np.testing.assert_allclose(out.numpy(), g["full_pred"], atol=1e-6)


def test_camera_and_projection_match_reference_golden():
    from gaussianavatar_amd.camera import test_pose_camera
    from oracle import lbs_oracle as O
    g = np.load(os.path.join(GOLD, "camera_loss_golden.npz"))
    for size in (1024, 512, 256):
        cam = test_pose_camera(size)
        np.testing.assert_allclose(cam["world_view_transform"], g[f"wvt_{size}"], atol=1e-6)
        np.testing.assert_allclose(cam["full_proj_transform"], g[f"full_{size}"], atol=2e-6)
        np.testing.assert_allclose(cam["camera_center"], g[f"center_{size}"], atol=1e-6)
        np.testing.assert_allclose([cam["FovX"], cam["FovY"]], g[f"fov_{size}"], atol=1e-12)
    out = O.project_points(torch.tensor(g["proj_pts"]), torch.tensor(g["full_1024"]))
    np.testing.assert_allclose(out.numpy(), g["proj_out"], rtol=1e-5, atol=1e-6)


def test_losses_match_reference_golden():
    from gaussianavatar_amd.losses import l1_loss_w, ssim
    g = np.load(os.path.join(GOLD, "camera_loss_golden.npz"))
    a, b = torch.tensor(g["loss_a"]), torch.tensor(g["loss_b"])
    assert abs(float(l1_loss_w(a, b)) - float(g["l1"])) < 1e-7
    assert abs(float(ssim(a, b)) - float(g["ssim"])) < 1e-6


def test_network_matches_reference_golden():
    """State-dict compatibility and numerical identity with the reference's POP_no_unet /
    UnetNoCond5DS (train mode, batch statistics)."""
    from gaussianavatar_amd.network import POP_no_unet, UnetNoCond5DS
    n = np.load(os.path.join(GOLD, "net_golden.npz"))
    sd = {k[4:]: torch.tensor(n[k]) for k in n.files if k.startswith("net.")}
    net = POP_no_unet(c_geom=8, geom_layer_type="conv", nf=4, hsize=16)
    net.train()
    net.load_state_dict(sd, strict=True)
    r, s, c = net(None, torch.tensor(n["geom"]), torch.tensor(n["uv"]))
    for got, key in ((r, "res1"), (s, "scales1"), (c, "shs1")):
        np.testing.assert_allclose(got.detach().numpy(), n[key], atol=2e-5)
    net.load_state_dict(sd, strict=True)
    r, s, c = net(torch.tensor(n["posef"]), torch.tensor(n["geom"]), torch.tensor(n["uv"]))
    for got, key in ((r, "res2"), (s, "scales2"), (c, "shs2")):
        np.testing.assert_allclose(got.detach().numpy(), n[key], atol=2e-5)
    u = UnetNoCond5DS(3, 8, 4)
    u.train()
    u.load_state_dict({k[5:]: torch.tensor(n[k]) for k in n.files if k.startswith("unet.")}, strict=True)
    np.testing.assert_allclose(u(torch.tensor(n["unet_x"])).detach().numpy(), n["unet_y"], atol=1e-5)


def test_stage1_dedup_equals_batched_evaluation():
    """Evaluating the decoder once and broadcasting == the reference's B expanded copies,
    values and gradients."""
    from gaussianavatar_amd.network import POP_no_unet
    torch.manual_seed(0)
    net = POP_no_unet(c_geom=8, hsize=16)
    net.train()
    geo = torch.randn(1, 8, 16, 16, requires_grad=True)
    idx = torch.stack(torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij"), -1).reshape(-1, 2).float() / 31
    B = 3
    w = [torch.randn(B, 1024, k) for k in (3, 1, 3)]
    outs = net.forward_points(None, geo.expand(B, -1, -1, -1), idx[None].expand(B, -1, -1), dedup=True)
    loss = sum((o * wi).sum() for o, wi in zip(outs, w))
    g1 = torch.autograd.grad(loss, [geo] + list(net.parameters()))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    outs2 = net.forward_points(None, geo.expand(B, -1, -1, -1).contiguous(), idx[None].expand(B, -1, -1).contiguous(), dedup=False)
    loss2 = sum((o * wi).sum() for o, wi in zip(outs2, w))
    g2 = torch.autograd.grad(loss2, [geo] + list(net.parameters()))
    for a, b in zip(outs, outs2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    # biases in front of a BatchNorm have an exactly-zero gradient in exact arithmetic; both
    # evaluations return float32 cancellation noise there, so the tolerance is absolute and
    # scaled by the largest gradient in the net
    gmax = max(float(t.abs().max()) for t in g2)
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, gmax), (float((a - b).abs().max()), gmax)


@pytest.mark.parametrize("kind", ["general", "avatar"])
def test_raster_oracle_matches_independent_torch_restatement(raster_oracle, kind):
    from oracle.raster_torch import rasterize
    sc = random_scene(250, 48, 32, seed=11, kind=kind, scale_med=0.05)
    st = raster_oracle.forward(sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], **cam_kwargs(sc))
    t = {k: torch.tensor(sc[k], requires_grad=True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    kw = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in cam_kwargs(sc).items()}
    color, aux = rasterize(t["means3D"], t["colors"], t["opacities"], t["scales"], t["rotations"], **kw)
    np.testing.assert_array_equal(aux["radii"].numpy(), st["radii"])
    np.testing.assert_array_equal(aux["rect"].numpy(), st["rect"])
    np.testing.assert_array_equal(aux["n_contrib"].numpy(), st["n_contrib"].astype(np.int32))
    assert np.abs(color.detach().numpy() - st["color"]).max() < 1e-5
    g = np.random.default_rng(5).normal(0, 1, (3, 32, 48)).astype(np.float32)
    (color * torch.tensor(g)).sum().backward()
    bo = raster_oracle.backward(st, g)
    for k, kk in (("means3D", "dmeans3D"), ("colors", "dcolors"), ("opacities", "dopacity"),
                  ("scales", "dscales"), ("rotations", "drots")):
        a, b = t[k].grad.numpy(), bo[kk]
        assert np.abs(a - b).max() <= 2e-5 * max(np.abs(a).max(), 1e-6), k


def test_raster_oracle_f64_gradients_match_finite_differences(raster_oracle_f64):
    O = raster_oracle_f64
    sc = random_scene(40, 32, 32, seed=2, kind="general", scale_med=0.08)
    g = np.random.default_rng(1).normal(0, 1, (3, 32, 32))

    def loss(means, scales):
        st = O.forward(means, sc["colors"], sc["opacities"], scales, sc["rotations"], **cam_kwargs(sc))
        return float((st["color"] * g).sum()), st

    l0, st = loss(sc["means3D"].astype(np.float64), sc["scales"].astype(np.float64))
    b = O.backward(st, g)
    rng = np.random.default_rng(0)
    vis = np.nonzero(st["radii"] > 0)[0]
    eps = 1e-6
    for i in rng.choice(vis, 6, replace=False):
        for arr, key in ((sc["means3D"], "dmeans3D"), (sc["scales"], "dscales")):
            a = arr.astype(np.float64)
            for c in range(3):
                ap, am = a.copy(), a.copy()
                ap[i, c] += eps
                am[i, c] -= eps
                args = (ap, sc["scales"].astype(np.float64)) if key == "dmeans3D" else (sc["means3D"].astype(np.float64), ap)
                args_m = (am, sc["scales"].astype(np.float64)) if key == "dmeans3D" else (sc["means3D"].astype(np.float64), am)
                fd = (loss(*args)[0] - loss(*args_m)[0]) / (2 * eps)
                an = b[key][i, c]
                # straight-through alpha clamp / termination gates make a few directions non-smooth
                assert abs(fd - an) <= 5e-3 * max(abs(an), abs(fd), 1e-3) or abs(fd - an) < 1e-4, (key, i, c, fd, an)


def test_raster_oracle_regression_vectors(raster_oracle):
    g = np.load(os.path.join(GOLD, "raster_golden.npz"))
    for name in ("gen", "ava"):
        st = raster_oracle.forward(g[f"{name}_means3D"], g[f"{name}_colors"], g[f"{name}_opacities"],
                                   g[f"{name}_scales"], g[f"{name}_rotations"],
                                   viewmatrix=g[f"{name}_viewmatrix"], projmatrix=g[f"{name}_projmatrix"],
                                   bg=g[f"{name}_bg"], W=48, H=32, tanfovx=float(g[f"{name}_tan"][0]),
                                   tanfovy=float(g[f"{name}_tan"][1]))
        for k in ("radii", "rect", "tiles_touched", "ranges", "point_list", "n_contrib"):
            np.testing.assert_array_equal(st[k], g[f"{name}_{k}"])
        np.testing.assert_allclose(st["color"], g[f"{name}_color"], atol=1e-6)
        b = raster_oracle.backward(st, g[f"{name}_g"])
        for k in ("dmeans3D", "dcolors", "dopacity", "dscales", "drots"):
            np.testing.assert_allclose(b[k], g[f"{name}_{k}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_oracle_matches_autograd_restatement(raster_oracle_f64, deg):
    """SH colour path (API completeness; the reference itself always passes colors_precomp):
    analytic C backward vs autograd of the independent torch formulation, float64."""
    import torch
    from oracle import raster_torch
    rng = np.random.default_rng(deg)
    P, M = 300, 16
    means = rng.normal(0, 1, (P, 3))
    campos = np.array([0.1, 0.3, 2.5])
    shs = rng.normal(0, 2.0, (P, M, 3))
    col, clamped = raster_oracle_f64.sh_colors(means, shs, deg, campos)
    assert 0.05 < clamped.mean() < 0.6          # both branches of the clamp are exercised
    mt = torch.tensor(means, dtype=torch.float64, requires_grad=True)
    st = torch.tensor(shs, dtype=torch.float64, requires_grad=True)
    ct = raster_torch.sh_colors(mt, st, deg, torch.tensor(campos))
    np.testing.assert_allclose(col, ct.detach().numpy(), rtol=1e-12, atol=1e-12)
    g = rng.normal(0, 1, (P, 3))
    ct.backward(torch.tensor(g))
    dsh, dmean = raster_oracle_f64.sh_backward(means, shs, deg, campos, clamped, g)
    np.testing.assert_allclose(dsh, st.grad.numpy(), rtol=1e-10, atol=1e-12)
    mg = mt.grad.numpy() if mt.grad is not None else np.zeros((P, 3))   # degree 0 is view-independent
    np.testing.assert_allclose(dmean, mg, rtol=1e-9, atol=1e-11)


def _net_full(device="cpu"):
    from gaussianavatar_amd.network import POP_no_unet, UnetNoCond5DS
    from oracle.make_golden import fill_state, net_full_inputs
    net = fill_state(POP_no_unet(c_geom=64, geom_layer_type="conv", nf=32, hsize=128), seed=21).to(device)
    unet = fill_state(UnetNoCond5DS(3, 64, 32), seed=23).to(device)
    net.train()
    unet.train()
    inp = net_full_inputs()
    mv = lambda v: [t.to(device) for t in v] if isinstance(v, list) else (v.to(device) if torch.is_tensor(v) else v)
    return net, unet, {k: mv(v) for k, v in inp.items()}, np.load(os.path.join(GOLD, "net_full_golden.npz"))


def check_net_full(device, tol_out=2e-5, tol_grad=2e-4):
    """POP_no_unet / UnetNoCond5DS at the reference's production widths (c_geom 64, hsize 128, nf 32) against
    outputs AND gradients of the reference's own modules (oracle/make_golden.py:make_net_full). On a HIP
    device these widths take the fused MFMA decoder + fused up-sampling path."""
    from oracle.make_golden import NET_FULL_KEEP
    net, unet, inp, n = _net_full(device)
    B = inp["B"]
    rel = lambda got, ref: float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
    for tag in ("s1", "s2"):
        geom = inp["geom"].clone().requires_grad_(True)
        posef = inp["posef"].clone().requires_grad_(True)
        net.zero_grad()
        outs = net(posef if tag == "s2" else None, geom.expand(B, -1, -1, -1).contiguous(), inp["uv"])
        for o, key in zip(outs, ("res", "scales", "shs")):
            assert rel(o.detach().cpu().numpy(), n[f"{tag}_{key}"]) <= tol_out, (tag, key)
        sum((o * w).sum() for o, w in zip(outs, inp["w"])).backward()
        assert rel(geom.grad[:, ::2].cpu().numpy(), n[f"{tag}_dgeom"]) <= tol_grad, tag
        if tag == "s2":
            assert rel(posef.grad[:, ::4].cpu().numpy(), n["s2_dposef"]) <= tol_grad
        params = dict(net.named_parameters())
        # a conv bias in front of BatchNorm has zero gradient in exact arithmetic: compare those on the
        # scale of the layer's weight gradient instead of their own (noise) scale
        for k in NET_FULL_KEEP:
            g = params[k].grad.detach().cpu().numpy()
            g = g[:8] if k.startswith("geom_proc") else g
            ref = n[f"{tag}_d.{k}"]
            scale = np.abs(ref).max()
            if k.endswith("conv4.bias"):
                scale = max(scale, float(params["decoder.conv4.weight"].grad.abs().max()))
            assert np.abs(g - ref).max() <= tol_grad * scale, (tag, k, np.abs(g - ref).max(), scale)
    y = unet(inp["unet_x"].clone())
    assert rel(y[:, ::8].detach().cpu().numpy(), n["unet_y"]) <= tol_out
    (y[:, ::8] * inp["unet_wy"]).sum().backward()
    up = dict(unet.named_parameters())
    for k in n.files:
        if k.startswith("unet_d."):
            g = up[k[7:]].grad.detach().cpu().numpy()
            assert np.abs(g - n[k]).max() <= tol_grad * max(np.abs(n[k]).max(), 1e-3), k


def test_production_width_nets_match_reference_outputs_and_grads():
    check_net_full("cpu")


@pytest.mark.parametrize("kind", ["avatar", "general"])
def test_fma_contraction_sensitivity(kind):
    """What "bit-exact against upstream" can mean. The un-vendored upstream is built by nvcc with FMA contraction
    on (-fmad=true); this repository's oracle and its HIP preprocess kernel are built with contraction OFF so that
    both follow one operation order. The measurable bound on the difference: build the SAME oracle source with
    contraction allowed (oracle/Makefile: -ffp-contract=fast -mfma, 185 fused instructions) and count the integer
    decisions that change at the headline size — radii (ceil of 3 sigma), tile rectangles ((int) casts), and with
    them tiles_touched and the pair count D. Measured in round 4: 0 of 200,000 Gaussians in both scenes, image mean
    |diff| 5e-7. The bars below leave room for a handful of flips on another compiler (a flip moves one Gaussian's
    radius by one pixel); DESIGN.md section 2 records the measured numbers."""
    from oracle.gsr_oracle import RasterOracle
    sc = random_scene(200_000, 1024, 1024, seed=1, kind=kind, scale_med=0.0035, spread=0.45)
    args = (sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    off = RasterOracle().forward(*args, **cam_kwargs(sc))
    on = RasterOracle(fma=True).forward(*args, **cam_kwargs(sc))
    radii = int((off["radii"] != on["radii"]).sum())
    rect = int((off["rect"] != on["rect"]).any(1).sum())
    print(f"\n{kind}: radii differing {radii}, rects {rect} of 200000; D {off['D']} vs {on['D']}; "
          f"image mean |diff| {np.abs(off['color'] - on['color']).mean():.2e}")
    assert radii <= 5 and rect <= 5
    assert abs(int(off["D"]) - int(on["D"])) <= 16
    assert np.abs(off["color"] - on["color"]).mean() <= 1e-5
    if radii == 0 and rect == 0:
        np.testing.assert_array_equal(off["point_list"], on["point_list"])      # then the sorted tile lists agree too
