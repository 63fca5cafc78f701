"""SMPL linear-blend skinning on the hot path, on top of the HIP library (include/galbs.h).

Mirrors what the reference reaches through `self.smpl_model.forward(...).A`
(/root/reference/submodules/smplx/body_models.py:306-393 -> lbs.py:152-252), the
`cano2live = A @ inv_mats` product (/root/reference/model/avatar_model.py:296) and the two
skinning einsums (/root/reference/model/avatar_model.py:311-314).
"""
from __future__ import annotations

import ctypes
from typing import NamedTuple, Optional

import torch

from . import _native


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(_native.raw_stream(device))


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensors must live on a HIP device (there is no CPU fallback)")


class _JointTransforms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, transl, joints_rest, parents, inv_mats):
        _require_cuda(pose, "joint_transforms")
        lib = _native.galbs()
        B = pose.shape[0]
        J = joints_rest.shape[0]
        ctx.set_materialize_grads(False)      # the loop uses only M: no zero-filled dA per iteration
        pose_c = pose.contiguous().float()
        transl_c = transl.contiguous().float() if transl is not None else None
        jr = joints_rest.contiguous().float()
        inv = inv_mats.float()
        if inv.dim() == 4:
            if inv.stride(0) == 0 or inv.shape[0] == 1:
                inv, inv_stride = inv[0].contiguous(), 0          # expanded [B,J,4,4] (avatar_model.py:89)
            else:
                inv, inv_stride = inv.contiguous(), J * 16
        else:
            inv, inv_stride = inv.contiguous(), 0
        A = torch.empty(B, J, 4, 4, dtype=torch.float32, device=pose.device)
        M = torch.empty_like(A)
        saved = torch.empty(B, lib.galbs_joint_saved_floats(J), dtype=torch.float32, device=pose.device)
        _native.galbs_check(lib.galbs_joint_transforms_fwd(
            B, J, _ptr(pose_c), _ptr(transl_c), _ptr(jr), _ptr(parents), _ptr(inv), inv_stride,
            _ptr(A), _ptr(M), _ptr(saved), _stream(pose.device)))
        ctx.save_for_backward(pose_c, jr, parents, inv, saved)
        ctx.inv_stride = inv_stride
        ctx.has_transl = transl is not None
        return A, M

    @staticmethod
    def backward(ctx, dA, dM):
        lib = _native.galbs()
        pose_c, jr, parents, inv, saved = ctx.saved_tensors
        B, J = pose_c.shape[0], jr.shape[0]
        dA = dA.contiguous().float() if dA is not None else None
        dM = dM.contiguous().float() if dM is not None else None
        dpose = torch.empty_like(pose_c) if ctx.needs_input_grad[0] else None
        dtransl = torch.empty(B, 3, dtype=torch.float32, device=pose_c.device) \
            if (ctx.has_transl and ctx.needs_input_grad[1]) else None
        if dpose is None and dtransl is None:
            return None, None, None, None, None
        _native.galbs_check(lib.galbs_joint_transforms_bwd(
            B, J, _ptr(pose_c), _ptr(jr), _ptr(parents), _ptr(inv), ctx.inv_stride, _ptr(saved),
            _ptr(dM), _ptr(dA), _ptr(dpose), _ptr(dtransl), _stream(pose_c.device)))
        return dpose, dtransl, None, None, None


def joint_transforms(pose, transl, joints_rest, parents, inv_mats):
    """pose [B,J*3] (axis-angle, joint 0 = global orientation), transl [B,3] or None,
    joints_rest [J,3], parents [J] int32, inv_mats [J,4,4] or [B,J,4,4]
    -> (A [B,J,4,4] incl. transl, cano2live [B,J,4,4] = A @ inv_mats). Differentiable w.r.t.
    pose and transl."""
    return _JointTransforms.apply(pose, transl, joints_rest, parents, inv_mats)


def _batched(t: torch.Tensor, inner: int):
    """Returns (contiguous tensor, batch stride in elements) treating an expanded leading
    dimension (stride 0, as the reference creates with .expand) as shared."""
    if t.dim() == 2:
        return t.contiguous().float(), 0
    if t.shape[0] == 1 or t.stride(0) == 0:
        return t[0].contiguous().float(), 0
    return t.contiguous().float(), inner


class _SkinPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, res, weights, mats):
        _require_cuda(mats, "skin")
        lib = _native.galbs()
        B, J = mats.shape[0], mats.shape[1]
        N = points.shape[-2]
        pts, pts_s = _batched(points, N * 3)
        rs, rs_s = _batched(res, N * 3) if res is not None else (None, 0)
        w, w_s = _batched(weights, N * J)
        m = mats.contiguous().float()
        out = torch.empty(B, N, 3, dtype=torch.float32, device=mats.device)
        _native.galbs_check(lib.galbs_skin_fwd(B, N, J, _ptr(pts), pts_s, _ptr(rs), rs_s, _ptr(w), w_s,
                                               _ptr(m), _ptr(out), _stream(mats.device)))
        ctx.save_for_backward(pts, rs if rs is not None else torch.empty(0, device=mats.device), w, m)
        ctx.meta = (B, N, J, pts_s, rs_s, w_s, res is not None,
                    None if res is None else tuple(res.shape), tuple(points.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _native.galbs()
        pts, rs, w, m = ctx.saved_tensors
        B, N, J, pts_s, rs_s, w_s, has_res, res_shape, pts_shape = ctx.meta
        if not has_res:
            rs = None
        need_x = ctx.needs_input_grad[0] or (has_res and ctx.needs_input_grad[1])
        need_m = ctx.needs_input_grad[3]
        dout = dout.contiguous().float()
        dres = torch.empty(B, N, 3, dtype=torch.float32, device=m.device) if need_x else None
        dmats = torch.empty(B, J, 4, 4, dtype=torch.float32, device=m.device) if need_m else None
        _native.galbs_check(lib.galbs_skin_bwd(B, N, J, _ptr(pts), pts_s, _ptr(rs), rs_s, _ptr(w), w_s,
                                               _ptr(m), _ptr(dout), _ptr(dres), _ptr(dmats),
                                               _stream(m.device)))

        def reduce_to(shape):
            if dres is None:
                return None
            g = dres
            if len(shape) == 2 or shape[0] == 1:
                g = g.sum(0, keepdim=len(shape) == 3)
            return g

        d_points = reduce_to(pts_shape) if ctx.needs_input_grad[0] else None
        d_res = reduce_to(res_shape) if (has_res and ctx.needs_input_grad[1]) else None
        return d_points, d_res, None, dmats


def skin(points, res, weights, mats):
    """Fused `pt_mats = sum_j w_nj M_j ; out = R (points + res) + t`.
    points [N,3] | [B,N,3], res likewise or None, weights [N,J] | [B,N,J], mats [B,J,4,4]
    -> [B,N,3]. Differentiable w.r.t. points, res and mats. Expanded (stride-0) batch
    dimensions are read once."""
    return _SkinPoints.apply(points, res, weights, mats)


class BodyOutput(NamedTuple):
    """The fields of smplx's SMPLOutput that the hot path consumes."""
    A: torch.Tensor
    cano2live: Optional[torch.Tensor]
    global_orient: torch.Tensor
    body_pose: torch.Tensor
    betas: Optional[torch.Tensor]


class SMPLBody(torch.nn.Module):
    """Pose -> joint transforms for an SMPL-family body model (SMPL: 24 joints, SMPL-X: 55).

    Holds only what the render-and-fit path needs: the kinematic tree and the rest-pose
    joints J(betas). `from_shape_space` evaluates lbs.py:206-210 once (betas are constant
    during training, /root/reference/model/avatar_model.py:95-98); the vertex path of
    lbs() is not reproduced (its outputs are unused by the reference's hot path).
    """

    def __init__(self, joints_rest: torch.Tensor, parents):
        super().__init__()
        parents = torch.as_tensor(parents, dtype=torch.int32)
        assert parents[0] == -1 and all(int(parents[i]) < i for i in range(1, len(parents)))
        self.register_buffer("joints_rest", joints_rest.float().contiguous())
        self.register_buffer("parents", parents.contiguous())
        self.num_joints = int(parents.shape[0])

    @classmethod
    def from_shape_space(cls, betas, v_template, shapedirs, J_regressor, parents):
        """J = J_regressor . (v_template + shapedirs . betas)   (lbs.py:206-210)."""
        v_shaped = v_template + torch.einsum("l,mkl->mk", betas.reshape(-1).float(), shapedirs.float())
        J = torch.einsum("ik,ji->jk", v_shaped, J_regressor.float())
        return cls(J, parents)

    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, inv_mats=None,
                full_pose=None, **extra_pose):
        """Same keyword surface as smplx's SMPL.forward / SMPLX.forward for the arguments the
        reference passes (avatar_model.py:280-294). SMPL-X extra pose blocks (jaw_pose, leye_pose,
        reye_pose, left_hand_pose, right_hand_pose) are concatenated in the SMPL-X joint order
        (body_models.py:1240-1247)."""
        if full_pose is None:
            parts = [global_orient, body_pose]
            for k in ("jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose"):
                if extra_pose.get(k) is not None:
                    parts.append(extra_pose[k])
            full_pose = torch.cat([p.reshape(p.shape[0], -1) for p in parts], dim=1)
        # `full_pose` [B, 3J]: the already concatenated axis-angle vector (skips the split + cat of the
        # keyword form and their slice/zero-fill/add backward when the caller holds it in one tensor)
        assert full_pose.shape[1] == self.num_joints * 3, (full_pose.shape, self.num_joints)
        if inv_mats is None:
            inv_mats = torch.eye(4, device=full_pose.device).expand(self.num_joints, 4, 4)
        A, M = joint_transforms(full_pose, transl, self.joints_rest, self.parents, inv_mats)
        return BodyOutput(A=A, cano2live=M, global_orient=global_orient, body_pose=body_pose, betas=betas)
