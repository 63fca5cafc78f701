"""CPU ORACLE (test infrastructure, NOT product code) for the SMPL LBS part of the hot path.

Restates, in plain torch on the CPU, exactly what the reference computes on the way from pose
parameters to deformed points:

  rest joints      lbs.py:206-210   v_shaped = v_template + shapedirs . betas ; J = J_regressor . v_shaped
  rodrigues        lbs.py:299-333   angle = ||v + 1e-8||, R = I + sin K + (1 - cos) K^2
  kinematic chain  lbs.py:349-405   G_i = G_parent(i) . [R_i | j_i - j_parent(i)], A_i = G_i with
                                    its translation reduced by G_i[:3,:3] j_i
  transl           body_models.py:383   A[:, :, :3, 3] += transl
  cano2live        avatar_model.py:296  A @ inv_mats
  skinning         avatar_model.py:311-314  pt_mats = sum_j w_nj M_j ; x' = R (x + res) + t
  projection + L1  utils/graphics_utils.py:12-19, utils/loss_utils.py:7-8 (the CPU baseline path
                   of BASELINE.md §3)

PINNED: tests/golden/lbs_golden.npz and skin_golden.npz were produced by importing the
reference's own submodules/smplx/lbs.py in the build container (oracle/make_golden.py) and
tests/test_oracle_golden.py checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import torch


def rest_joints(betas, v_template, shapedirs, J_regressor):
    """[B,10] -> J [B,Jn,3] (lbs.py:206-210)."""
    v_shaped = v_template[None] + torch.einsum("bl,mkl->bmk", betas, shapedirs)
    return torch.einsum("bik,ji->bjk", v_shaped, J_regressor)


def rodrigues(rot_vecs):
    """[N,3] axis-angle -> [N,3,3] (lbs.py:299-333)."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    d = rot_vecs / angle
    c = torch.cos(angle)[:, :, None]
    s = torch.sin(angle)[:, :, None]
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = torch.zeros_like(rx)
    K = torch.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).reshape(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return eye + s * K + (1 - c) * torch.bmm(K, K)


def joint_transforms(pose, transl, J, parents):
    """pose [B,Jn*3], transl [B,3] or None, J [B,Jn,3] or [Jn,3], parents [Jn] -> A [B,Jn,4,4]
    (the SMPLOutput.A of the reference, translation included)."""
    B = pose.shape[0]
    Jn = parents.shape[0]
    if J.dim() == 2:
        J = J[None].expand(B, -1, -1)
    R = rodrigues(pose.reshape(-1, 3)).reshape(B, Jn, 3, 3)
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T = torch.zeros(B, Jn, 4, 4, dtype=pose.dtype)
    T[:, :, :3, :3] = R
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, Jn):
        chain.append(chain[int(parents[i])] @ T[:, i])
    G = torch.stack(chain, 1)
    Jh = torch.cat([J, torch.zeros(B, Jn, 1, dtype=pose.dtype)], 2)[..., None]    # [B,Jn,4,1]
    corr = G @ Jh                                                                   # [B,Jn,4,1]
    A = G.clone()
    A[:, :, :, 3] = G[:, :, :, 3] - corr[..., 0]
    if transl is not None:
        A[:, :, :3, 3] = A[:, :, :3, 3] + transl[:, None]
    return A


def lbs_full(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """The WHOLE of the reference's lbs() as `SMPL.forward` runs it every iteration (lbs.py:206-247; called at
    body_models.py:369) — including what the render-and-fit path never reads: the pose blend shapes (lbs.py:216-229),
    the per-vertex blend of the joint transforms and the vertex skinning (:239-247). betas [B,10], pose [B,Jn*3] ->
    (verts [B,V,3], joints [B,Jn,3], A [B,Jn,4,4] without transl). bench.py's cpu_baseline times this form next to
    `joint_transforms`, which is the same with the dead work removed."""
    B = pose.shape[0]
    Jn = parents.shape[0]
    v_shaped = v_template[None] + torch.einsum("bl,mkl->bmk", betas, shapedirs)            # :206-207
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)                                  # :210
    rot = rodrigues(pose.reshape(-1, 3)).reshape(B, Jn, 3, 3)                               # :216-218
    pose_feature = (rot[:, 1:] - torch.eye(3, dtype=pose.dtype)).reshape(B, -1)             # :220
    v_posed = torch.matmul(pose_feature, posedirs).reshape(B, -1, 3) + v_shaped             # :222-232
    A = joint_transforms(pose, None, J, parents)                                            # :234 (batch_rigid_transform)
    joints = (A @ torch.cat([J, torch.ones(B, Jn, 1, dtype=pose.dtype)], 2)[..., None])[..., :3, 0]
    T = torch.matmul(lbs_weights[None].expand(B, -1, -1), A.reshape(B, Jn, 16)).reshape(B, -1, 4, 4)   # :238-241
    v_homo = torch.matmul(T, torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=pose.dtype)], 2)[..., None])
    return v_homo[:, :, :3, 0], joints, A                                                   # :243-247


def cano2live(A, inv_mats):
    return A @ inv_mats


def skin(query_points, res, weights, mats):
    """query_points [B,N,3], res [B,N,3], weights [B,N,J], mats [B,J,4,4] -> [B,N,3]
    (avatar_model.py:308-314)."""
    pt = torch.einsum("bnj,bjxy->bnxy", weights, mats)
    x = query_points + res
    return torch.einsum("bnxy,bny->bnx", pt[..., :3, :3], x) + pt[..., :3, 3]


def project_points(points, full_proj_transform):
    """geom_transform_points (utils/graphics_utils.py:12-19): [P,3] x [4,4] -> [P,3]."""
    ones = torch.ones(points.shape[0], 1, dtype=points.dtype)
    out = torch.cat([points, ones], 1) @ full_proj_transform
    return out[:, :3] / (out[:, 3:] + 0.0000001)


def cpu_baseline_step(pose, transl, J, parents, inv_mats, query_points, res, weights,
                      full_proj_transform, body=None, betas=None):
    """One fwd+bwd of the reference's PyTorch-CPU LBS + projection path with an L1-to-black
    loss. pose/res must require grad. Returns the loss.
    body = dict(v_template, shapedirs, posedirs, J_regressor, lbs_weights) + betas: the body model evaluated AS THE
    REFERENCE RUNS IT — the whole lbs() with its 6,890-vertex blend shapes and vertex skinning, whose vertices the path
    discards (`lbs_full`); body = None: rest joints precomputed, joint transforms only (the dead work removed)."""
    if body is not None:
        _verts, _joints, A = lbs_full(betas, pose, body["v_template"], body["shapedirs"], body["posedirs"],
                                      body["J_regressor"], parents, body["lbs_weights"])
        A = A.clone()
        A[:, :, :3, 3] = A[:, :, :3, 3] + transl[:, None]                                   # body_models.py:383
    else:
        A = joint_transforms(pose, transl, J, parents)
    M = cano2live(A, inv_mats)
    full = skin(query_points, res, weights, M)
    loss = 0.0
    for b in range(full.shape[0]):
        loss = loss + torch.abs(project_points(full[b], full_proj_transform)).mean()
    loss.backward()
    return loss
