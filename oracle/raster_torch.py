"""Second, independently written restatement of the rasterizer semantics (SURVEY.md
Appendix A) as dense, vectorised torch code whose gradients come from autograd.

TEST INFRASTRUCTURE ONLY (oracle/). PARITY UNPINNED — see oracle/gsr_oracle.c.

Purpose: cross-check oracle/gsr_oracle.c (sequential loops + hand-derived analytic backward)
with a different formulation (global sort + masks + cumprod, autograd backward) on small
scenes. The three places where a naive autograd graph would diverge from the analytic
backward of the CUDA reference (Appendix A.5b) are patched explicitly:
  (i)   alpha = min(0.99, o*G) is straight-through in the backward pass,
  (ii)  the frustum clamp of t.x/t.z, t.y/t.z zeroes d/dt.x (d/dt.y) and ignores du/dt.z,
  (iii) the conic backward uses 1/(det^2 + 1e-7).
Memory is O(pixels x Gaussians): keep P <= ~2000 and images <= 64x64.
"""
from __future__ import annotations

import torch

TILE = 16


class _ConicFromCov(torch.autograd.Function):
    """(a, b, c) -> (A, B, C) = (c, -b, a)/det with the reference's epsilon-ed backward."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c, det)
        inv = 1.0 / det
        return c * inv, -b * inv, a * inv

    @staticmethod
    def backward(ctx, gA, gB, gC):
        a, b, c, det = ctx.saved_tensors
        k = 1.0 / (det * det + 1e-7)
        da = k * (-c * c * gA + b * c * gB + (det - a * c) * gC)
        dc = k * (-a * a * gC + a * b * gB + (det - a * c) * gA)
        db = k * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC)
        return da, db, dc


def _rotmat(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def rasterize(means3D, colors, opacities, scales, rotations, *, viewmatrix, projmatrix, bg,
              W, H, tanfovx, tanfovy, scale_modifier=1.0, cov3D_precomp=None):
    """Returns (color[3,H,W], aux dict). Differentiable w.r.t. means3D, colors, opacities,
    scales, rotations (or cov3D_precomp)."""
    dt = means3D.dtype
    P = means3D.shape[0]
    V = viewmatrix.to(dt).reshape(4, 4).t()    # column-vector form
    PV = projmatrix.to(dt).reshape(4, 4).t()
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    tfx = torch.tensor(tanfovx, dtype=dt)
    tfy = torch.tensor(tanfovy, dtype=dt)
    fx = torch.tensor(float(W), dtype=dt) / (2.0 * tfx)
    fy = torch.tensor(float(H), dtype=dt) / (2.0 * tfy)
    # Every scalar expression below is written out in the operation order of
    # oracle/gsr_oracle.c so that the float32 integer-determining stage agrees bit for bit.
    X, Y, Z = means3D.unbind(-1)

    def row(Mx, r):
        return Mx[r, 0] * X + Mx[r, 1] * Y + Mx[r, 2] * Z + Mx[r, 3]

    tx, ty, tz = row(V, 0), row(V, 1), row(V, 2)
    hx, hy, hw = row(PV, 0), row(PV, 1), row(PV, 3)
    vis = tz > 0.2
    winv = 1.0 / (hw + 1e-7)
    ndc = torch.stack([hx * winv, hy * winv], 1)
    if cov3D_precomp is None:
        Rm = _rotmat(rotations)
        s = scale_modifier * scales
        M = [[s[:, k] * Rm[:, a_, k] for a_ in range(3)] for k in range(3)]   # M[k][a]
        Sg = [[M[0][a_] * M[0][b_] + M[1][a_] * M[1][b_] + M[2][a_] * M[2][b_]
               for b_ in range(3)] for a_ in range(3)]
    else:
        c6 = cov3D_precomp
        Sg = [[c6[:, 0], c6[:, 1], c6[:, 2]], [c6[:, 1], c6[:, 3], c6[:, 4]],
              [c6[:, 2], c6[:, 4], c6[:, 5]]]
    limx, limy = 1.3 * tfx, 1.3 * tfy
    tzs = torch.where(vis, tz, torch.ones_like(tz))
    rx, ry = tx / tzs, ty / tzs
    clx = (rx < -limx) | (rx > limx)
    cly = (ry < -limy) | (ry > limy)
    # patch (ii): value = clamp(t.x/t.z)*t.z as in the forward pass; gradient = d/dt.x of t.x
    # when unclamped, zero when clamped, and no dependence on t.z either way
    uval = (torch.minimum(limx, torch.maximum(-limx, rx)) * tzs).detach()
    vval = (torch.minimum(limy, torch.maximum(-limy, ry)) * tzs).detach()
    u = torch.where(clx, uval, tx + (uval - tx).detach())
    v = torch.where(cly, vval, ty + (vval - ty).detach())
    J00, J02 = fx / tzs, -(fx * u) / (tzs * tzs)
    J11, J12 = fy / tzs, -(fy * v) / (tzs * tzs)
    T0 = [J00 * V[0, j] + J02 * V[2, j] for j in range(3)]
    T1 = [J11 * V[1, j] + J12 * V[2, j] for j in range(3)]
    s0 = [Sg[p][0] * T0[0] + Sg[p][1] * T0[1] + Sg[p][2] * T0[2] for p in range(3)]
    s1 = [Sg[p][0] * T1[0] + Sg[p][1] * T1[1] + Sg[p][2] * T1[2] for p in range(3)]
    a = (T0[0] * s0[0] + T0[1] * s0[1] + T0[2] * s0[2]) + 0.3
    b = T1[0] * s0[0] + T1[1] * s0[1] + T1[2] * s0[2]
    c = (T1[0] * s1[0] + T1[1] * s1[1] + T1[2] * s1[2]) + 0.3
    det = a * c - b * b
    vis = vis & (det != 0)
    safe = lambda x, fill: torch.where(vis, x, torch.full_like(x, fill))
    A, B, C = _ConicFromCov.apply(safe(a, 1.0), safe(b, 0.0), safe(c, 1.0))
    mid = 0.5 * (a + c)
    sq = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    lam = torch.maximum(mid + sq, mid - sq)
    rad = torch.ceil(3.0 * torch.sqrt(lam))
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    def tc(val, hi):
        val = torch.nan_to_num(val.detach(), nan=0.0, posinf=1e9, neginf=-1e9)
        return torch.trunc(val).clamp(0, hi).to(torch.int64)

    x0, x1 = tc((px - rad) / TILE, gx), tc((px + rad + (TILE - 1)) / TILE, gx)
    y0, y1 = tc((py - rad) / TILE, gy), tc((py + rad + (TILE - 1)) / TILE, gy)
    tiles = (x1 - x0) * (y1 - y0)
    vis = vis & (tiles > 0)
    radii = torch.where(vis, rad.detach().clamp(max=2**31 - 1).to(torch.int64),
                        torch.zeros_like(tiles)).to(torch.int32)
    tiles = torch.where(vis, tiles, torch.zeros_like(tiles))
    rect = torch.stack([x0, y0, x1, y1], 1) * vis[:, None]

    # global order by (depth bits, index); per-tile lists are sub-sequences of it
    dbits = tz.detach().to(torch.float32).view(torch.int32).to(torch.int64)
    key = dbits * (1 << 32) + torch.arange(P)
    key = torch.where(vis, key, torch.full_like(key, torch.iinfo(torch.int64).max))
    order = torch.argsort(key)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxs, pys = xs.reshape(-1), ys.reshape(-1)
    ptx, pty = pxs // TILE, pys // TILE
    o = order
    in_tile = (vis[o][None] & (ptx[:, None] >= x0[o][None]) & (ptx[:, None] < x1[o][None]) &
               (pty[:, None] >= y0[o][None]) & (pty[:, None] < y1[o][None]))       # [Npix,P]
    dx = px[o][None] - pxs[:, None].to(dt)
    dy = py[o][None] - pys[:, None].to(dt)
    power = -0.5 * (A[o][None] * dx * dx + C[o][None] * dy * dy) - B[o][None] * dx * dy
    G = torch.exp(torch.where(in_tile, power, torch.zeros_like(power)).clamp(max=0))
    raw = opacities.reshape(-1)[o][None] * G
    alpha = raw + (raw.clamp(max=0.99) - raw).detach()          # patch (i): straight-through
    live = in_tile & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha = torch.where(live, alpha, torch.zeros_like(alpha))
    Tn = torch.cumprod(1.0 - alpha, 1)                             # T after each entry
    stop = live & (Tn.detach() < 1e-4)
    stopped = torch.cumsum(stop.to(torch.int64), 1) > 0            # entry at/after termination
    alpha = torch.where(stopped, torch.zeros_like(alpha), alpha)
    contrib = live & ~stopped
    Tn = torch.cumprod(1.0 - alpha, 1)
    Tb = torch.cat([torch.ones(Tn.shape[0], 1, dtype=dt), Tn[:, :-1]], 1)   # T before entry
    wgt = alpha * Tb
    col = wgt @ colors[o]                                          # [Npix,3]
    Tfin = Tn[:, -1] if P > 0 else torch.ones(H * W, dtype=dt)
    out = col + Tfin[:, None] * bg.to(dt)[None]
    color = out.t().reshape(3, H, W)
    pos = torch.cumsum(in_tile.to(torch.int64), 1)                 # 1-based position in tile list
    n_contrib = torch.where(contrib, pos, torch.zeros_like(pos)).max(1).values if P > 0 else \
        torch.zeros(H * W, dtype=torch.int64)
    aux = dict(radii=radii, rect=rect.to(torch.int32), tiles_touched=tiles.to(torch.int32),
               final_T=Tfin.detach(), n_contrib=n_contrib.to(torch.int32), order=order,
               xy=torch.stack([px, py], 1).detach(), depth=tz.detach(),
               conic=torch.stack([A, B, C], 1).detach(), visible=vis)
    return color, aux


# Real SH basis of the published 3DGS convention (Condon–Shortley signs folded in), as a list of
# monomial expressions: sh_colors() below is the autograd restatement of oracle/gsr_oracle.c's
# gsro_sh_forward / gsro_sh_backward.
def _sh_basis(x, y, z):
    xx, yy, zz = x * x, y * y, z * z
    one = torch.ones_like(x)
    return [
        0.28209479177387814 * one,
        -0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x,
        1.0925484305920792 * x * y, -1.0925484305920792 * y * z,
        0.31539156525252005 * (2 * zz - xx - yy), -1.0925484305920792 * x * z,
        0.5462742152960396 * (xx - yy),
        -0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * x * y * z,
        -0.4570457994644658 * y * (4 * zz - xx - yy), 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy),
        -0.4570457994644658 * x * (4 * zz - xx - yy), 1.445305721320277 * z * (xx - yy),
        -0.5900435899266435 * x * (xx - 3 * yy)]


def sh_colors(means3D, shs, sh_degree, campos):
    """colors [P,3] = max(0, sum_k basis_k(normalize(mean - campos)) * shs[:, k] + 0.5)."""
    d = means3D - campos.to(means3D.dtype).reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    basis = torch.stack(_sh_basis(d[:, 0], d[:, 1], d[:, 2])[:(sh_degree + 1) ** 2], dim=1)   # [P,K]
    raw = (basis.unsqueeze(-1) * shs[:, :basis.shape[1]]).sum(1) + 0.5
    return torch.clamp_min(raw, 0.0)
