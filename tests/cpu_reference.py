"""All-CPU evaluation of the reference's train_stage1 / train_stage2 formulas — TEST INFRASTRUCTURE.

The nets are the product modules on CPU (their plain-torch formulation, pinned to the reference's own
POP_no_unet / UnetNoCond5DS outputs by tests/golden/net_golden.npz), the body is oracle/lbs_oracle.py
(pinned by lbs_golden.npz), the rasterizer is oracle/gsr_oracle.c wrapped as an autograd function
(forward = C oracle forward, backward = its analytic backward). Layout and glue follow
/root/reference/model/avatar_model.py:277-368 (stage 1) and :369-463 (stage 2) line by line: B expanded
copies through the net, permutes, `* 0.02`, bool-mask gather, scale warm-up, `repeat(1,1,3)`, per-frame
render loop.
"""
from __future__ import annotations

import copy
import math

import numpy as np
import torch


class OracleRaster(torch.autograd.Function):
    """colour image [3,H,W] of one frame through oracle/gsr_oracle.c (float32), differentiable
    w.r.t. means3D, colours, scales (the three the avatar path needs)."""

    @staticmethod
    def forward(ctx, means3D, colors, scales, rotations, opacity, cam, oracle):
        st = oracle.forward(means3D.detach().numpy(), colors.detach().numpy(), opacity.detach().numpy().reshape(-1),
                            scales.detach().numpy(), rotations.detach().numpy(), **cam)
        ctx.st, ctx.oracle = st, oracle
        return torch.from_numpy(st["color"].astype(np.float32))

    @staticmethod
    def backward(ctx, g):
        out = ctx.oracle.backward(ctx.st, g.contiguous().numpy())
        t = lambda k: torch.from_numpy(out[k].astype(np.float32))
        return t("dmeans3D"), t("dcolors"), t("dscales"), None, None, None, None


def snapshot(model):
    """CPU copies of everything trainable / stateful of an AvatarModel, taken BEFORE a step."""
    snap = dict(net=copy.deepcopy(model.net).cpu(), geo=model.geo_feature.detach().cpu().clone().requires_grad_(True),
                pose=model.pose.weight.detach().cpu().clone().requires_grad_(True),
                transl=model.transl.weight.detach().cpu().clone().requires_grad_(True))
    if hasattr(model, "pose_encoder"):
        snap["pose_encoder"] = copy.deepcopy(model.pose_encoder).cpu()
    return snap


def camera_kwargs(batch, b):
    """Per-frame camera of a batch dict (python scalars, lists or tensors) -> oracle kwargs."""
    def scalar(v):
        v = v[b]
        return float(v.item() if torch.is_tensor(v) else v)
    mat = lambda k: batch[k][b].detach().cpu().numpy()
    return dict(viewmatrix=mat("world_view_transform"), projmatrix=mat("full_proj_transform"),
                bg=None, W=int(scalar(batch["width"])), H=int(scalar(batch["height"])),
                tanfovx=math.tan(scalar(batch["FovX"]) * 0.5), tanfovy=math.tan(scalar(batch["FovY"]) * 0.5))


def forward(model, snap, batch, iteration, oracle, stage=1, free=False):
    """-> dict(image [B,3,H,W], full_pred, offset_loss, scale_loss, geo_loss / pose_loss), all CPU
    tensors attached to the parameters in `snap` (call .backward() on any scalar of them)."""
    from oracle import lbs_oracle as O
    net = snap["net"]
    net.train()
    idx = batch["pose_idx"].cpu() if "pose_idx" in batch else None
    B = (idx.shape[0] if idx is not None else batch["pose_data"].shape[0])
    if free and stage == 1:
        pose, transl = batch["pose_data"].cpu().float(), batch["transl_data"].cpu().float()
    else:
        pose, transl = snap["pose"][idx], snap["transl"][idx]
    J = model.smpl_model.joints_rest.cpu()
    parents = model.smpl_model.parents.long().cpu()
    if model.model_parms.smpl_type == "smplx":
        rest = batch["rest_pose"].cpu().float()
        pose = torch.cat([pose[:, :66], rest], dim=1)          # global | body | jaw | eyes | hands
    A = O.joint_transforms(pose, transl, J, parents)
    cano2live = A @ model.inv_mats.cpu()[:1]
    geom = snap["geo"].expand(B, -1, -1, -1).contiguous()
    uv = model.uv_coord_map.cpu()[None].expand(B, -1, -1).contiguous()
    posef = None
    if stage == 2:
        enc = snap["pose_encoder"]
        enc.train()
        posef = enc(batch["inp_pos_map"].cpu().float())
        posef.retain_grad()          # (tests that re-evaluate the encoder's backward in float64 read it)
    res, sc, shs = net(posef, geom, uv)
    res = res.permute(0, 2, 1) * 0.02
    sc = sc.permute(0, 2, 1)
    shs = shs.permute(0, 2, 1)
    valid = model.valid_idx.cpu()
    point_res = res[:, valid].contiguous()
    qp = model.query_points.cpu()[:1].expand(B, -1, -1)
    w = model.query_lbs.cpu()[:1].expand(B, -1, -1)
    full = O.skin(qp, point_res, w, cano2live)
    if stage == 1 and iteration < 1000:
        sc = sc * 1e-3 * iteration
    scales = sc[:, valid].contiguous().repeat(1, 1, 3)
    cols = shs[:, valid].contiguous()
    out = dict(full_pred=full, offset_loss=torch.mean(res ** 2), scale_loss=torch.mean(sc[:, valid]))
    if stage == 1:
        out["geo_loss"] = torch.mean(snap["geo"] ** 2)
    else:
        out["pose_loss"] = torch.mean(posef ** 2)
        out["pose_featmap"] = posef
    rots = model.fix_rotation.cpu()
    opac = model.fix_opacity.cpu()
    bg = model.background.cpu().numpy()
    images, raster_inputs = [], []
    for b in range(B):
        cam = camera_kwargs(batch, b)
        cam["bg"] = bg
        images.append(OracleRaster.apply(full[b], cols[b], scales[b], rots, opac, cam, oracle))
        # what the rasterizer saw, for callers that re-render with the float64 oracle (coin-toss counts)
        raster_inputs.append((full[b].detach().numpy(), cols[b].detach().numpy(), opac.detach().numpy().reshape(-1),
                              scales[b].detach().numpy(), rots.detach().numpy(), cam))
    out["image"] = torch.stack(images)
    out["raster_inputs"] = raster_inputs
    return out


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
