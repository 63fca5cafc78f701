"""GPU parity: HIP LBS kernels (through include/galbs.h) vs the CPU oracle and the golden
vectors produced by the reference's own lbs.py (tests/golden/lbs_golden.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5          # float32, |values| <= ~3


def _golden():
    return np.load(os.path.join(GOLD, "lbs_golden.npz"))


@pytest.mark.parametrize("model", ["smpl", "smplx"])
def test_joint_transforms_match_reference_golden(model):
    from gaussianavatar_amd.lbs import SMPLBody
    g = _golden()
    t = lambda k: torch.tensor(g[f"{model}_{k}"])
    body = SMPLBody.from_shape_space(t("betas")[0], t("v_template"), t("shapedirs"), t("J_regressor"),
                                     g[f"{model}_parents"]).cuda()
    pose, transl = t("pose").cuda(), t("transl").cuda()
    out = body.forward(global_orient=pose[:, :3], body_pose=pose[:, 3:], transl=transl)
    np.testing.assert_allclose(out.A.cpu().numpy(), g[f"{model}_A"], rtol=0, atol=TOL)
    # identity inv_mats -> cano2live == A
    np.testing.assert_allclose(out.cano2live.cpu().numpy(), g[f"{model}_A"], rtol=0, atol=TOL)


def test_joint_transforms_backward_vs_oracle():
    from gaussianavatar_amd.lbs import joint_transforms
    from oracle import lbs_oracle as O
    g = _golden()
    torch.manual_seed(0)
    B, J = 3, 24
    parents = torch.tensor(g["smpl_parents"], dtype=torch.long)
    Jr = torch.randn(J, 3) * 0.3
    inv = torch.linalg.inv(O.joint_transforms(torch.randn(1, 72) * 0.2, torch.randn(1, 3), Jr, parents))[0]
    pose = (torch.tensor(g["smpl_pose"][1:4])).requires_grad_(True)
    transl = torch.randn(B, 3).requires_grad_(True)
    wA, wM = torch.randn(B, J, 4, 4), torch.randn(B, J, 4, 4)
    A = O.joint_transforms(pose, transl, Jr, parents)
    M = O.cano2live(A, inv[None].expand(B, -1, -1, -1))
    ((A * wA).sum() + (M * wM).sum()).backward()
    pg = pose.detach().cuda().requires_grad_(True)
    tg = transl.detach().cuda().requires_grad_(True)
    Ag, Mg = joint_transforms(pg, tg, Jr.cuda(), parents.int().cuda(), inv.cuda())
    ((Ag * wA.cuda()).sum() + (Mg * wM.cuda()).sum()).backward()
    np.testing.assert_allclose(Ag.detach().cpu().numpy(), A.detach().numpy(), atol=TOL)
    np.testing.assert_allclose(Mg.detach().cpu().numpy(), M.detach().numpy(), atol=5 * TOL)
    sc = float(pose.grad.abs().max())
    assert float((pg.grad.cpu() - pose.grad).abs().max()) <= 2e-4 * sc
    assert float((tg.grad.cpu() - transl.grad).abs().max()) <= 2e-4 * float(transl.grad.abs().max())


def test_zero_pose_rodrigues_is_finite():
    from gaussianavatar_amd.lbs import joint_transforms
    J = 24
    parents = torch.tensor(_golden()["smpl_parents"]).int().cuda()
    pose = torch.zeros(2, 72, device="cuda", requires_grad=True)
    A, M = joint_transforms(pose, None, torch.randn(J, 3).cuda(), parents, torch.eye(4).expand(J, 4, 4).cuda())
    (A.sum() + M.sum()).backward()
    assert torch.isfinite(A).all() and torch.isfinite(pose.grad).all()


def test_skin_matches_reference_golden():
    from gaussianavatar_amd.lbs import skin
    g = np.load(os.path.join(GOLD, "skin_golden.npz"))
    c = lambda k: torch.tensor(g[k]).cuda()
    out = skin(c("query_points"), c("res"), c("weights"), c("cano2live"))
    np.testing.assert_allclose(out.cpu().numpy(), g["full_pred"], rtol=0, atol=TOL)
    # shared (un-batched) points and weights give the same result
    out2 = skin(c("query_points")[0], c("res"), c("weights")[0], c("cano2live"))
    np.testing.assert_allclose(out2.cpu().numpy(), g["full_pred"], rtol=0, atol=TOL)


@pytest.mark.parametrize("B,N,J", [(2, 1000, 24), (3, 777, 55), (9, 300, 24), (2, 200_000, 24),
                                   (1, 250_000, 55)])   # last two: headline size, SMPL-X stage-2 size
def test_skin_backward_vs_oracle(B, N, J):
    from gaussianavatar_amd.lbs import skin
    from oracle import lbs_oracle as O
    torch.manual_seed(B * N)
    pts = torch.randn(N, 3) * 0.5
    res = (torch.randn(B, N, 3) * 0.02).requires_grad_(True)
    w = torch.rand(N, J) ** 6
    w[w < 0.02] = 0
    w = w / w.sum(1, keepdim=True).clamp(min=1e-8)
    mats = (torch.eye(4).expand(B, J, 4, 4) + 0.2 * torch.randn(B, J, 4, 4)).clone()
    mats[:, :, 3] = torch.tensor([0.0, 0, 0, 1])
    mats.requires_grad_(True)
    gout = torch.randn(B, N, 3)
    ref = O.skin(pts[None].expand(B, -1, -1), res, w[None].expand(B, -1, -1), mats)
    (ref * gout).sum().backward()
    rg = res.detach().cuda().requires_grad_(True)
    mg = mats.detach().cuda().requires_grad_(True)
    out = skin(pts.cuda(), rg, w.cuda(), mg)
    (out * gout.cuda()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=TOL)
    np.testing.assert_allclose(rg.grad.cpu().numpy(), res.grad.numpy(), atol=TOL)
    dm, dm_ref = mg.grad.cpu().numpy(), mats.grad.numpy()
    assert np.abs(dm[:, :, :3] - dm_ref[:, :, :3]).max() <= 1e-4 * np.abs(dm_ref).max()


def test_shared_residual_gradient_is_summed_over_frames():
    from gaussianavatar_amd.lbs import skin
    torch.manual_seed(1)
    B, N, J = 2, 500, 24
    pts = torch.randn(N, 3).cuda()
    w = torch.softmax(torch.randn(N, J) * 3, 1).cuda()
    mats = (torch.eye(4).expand(B, J, 4, 4) + 0.1 * torch.randn(B, J, 4, 4)).cuda()
    base = torch.randn(1, N, 3).cuda().requires_grad_(True)
    out = skin(pts, base.expand(B, -1, -1), w, mats)
    out.sum().backward()
    full = base.detach().expand(B, -1, -1).clone().requires_grad_(True)
    skin(pts, full, w, mats).sum().backward()
    torch.testing.assert_close(base.grad, full.grad.sum(0, keepdim=True), rtol=1e-5, atol=1e-5)
