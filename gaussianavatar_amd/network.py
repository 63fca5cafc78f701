"""Feature networks of the hot path: state-dict compatible torch modules whose HIP-device forward/backward run on
the hand-written kernels of csrc/ganet_*.hip (decoder MLP, 5x5 convolutions, up-sampling; gaussianavatar_amd/fused.py);
on the CPU — and for shapes the kernels do not cover, e.g. the stage-2 UNet — the plain torch formulation runs.

State-dict compatible re-implementation of the modules the reference instantiates for its
default flags (`geom_layer_type='conv'`, /root/reference/arguments/__init__.py:111):

  POP_no_unet      /root/reference/model/network.py:9-83
  GeomConvLayers   /root/reference/model/modules.py:114-137   3x conv5x5, no bias, no activation
  ShapeDecoder     /root/reference/model/modules.py:508-582   1x1-conv MLP + BN1d + softplus, 3 heads
  UnetNoCond5DS    /root/reference/model/modules.py:185-232   stage-2 pose encoder
  uv_to_grid       /root/reference/model/modules.py:745-754

Parameter/buffer names and shapes are identical, so reference checkpoints load
(`net.pth` keys, /root/reference/model/avatar_model.py:163-186). The computation is laid out
differently:
  * the decoder works point-major ([B*HW, C] row-major GEMMs instead of Conv1d over
    [B, C, HW]) — the layout rocBLAS likes and the one the downstream gather/skin wants;
  * `forward_points` returns [B, HW, C] directly (no permutes in the caller);
  * a stage-1 call with batch-invariant input (geo_feature.expand, pose_featmap=None) is
    evaluated ONCE and broadcast — BatchNorm batch statistics over B identical copies equal
    those of one copy, so outputs and gradients are unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused


def uv_to_grid(uv_idx_map: torch.Tensor, resolution: int) -> torch.Tensor:
    """[B, S*S, 2] in [0,1] -> grid_sample grid [B, S, S, 2] in [-1,1], with the reference's
    (row, col) -> (x, y) transpose (modules.py:745-754)."""
    bs = uv_idx_map.shape[0]
    grid = uv_idx_map.reshape(bs, resolution, resolution, 2) * 2 - 1.0
    return grid.transpose(1, 2)


class GeomConvLayers(nn.Module):
    def __init__(self, input_nc=16, hidden_nc=16, output_nc=16, use_relu=False):
        super().__init__()
        self.use_relu = use_relu
        chans = [(input_nc, hidden_nc), (hidden_nc, hidden_nc), (hidden_nc, output_nc)]
        for i, (ci, co) in enumerate(chans, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(ci, co, kernel_size=5, stride=1, padding=2, bias=False))

    def forward(self, x):
        weights = [getattr(self, f"conv{i}").weight for i in (1, 2, 3)]
        if not self.use_relu and fused.geom_convs_supported(x, weights):
            # hand-written channels-last kernels (csrc/ganet_conv.hip); the result is a logical-NCHW view of the
            # channels-last map the up-sampling kernel reads
            return fused.geom_convs(x, weights)
        for i in (1, 2, 3):
            x = getattr(self, f"conv{i}")(x)
            if self.use_relu and i < 3:
                x = F.leaky_relu(x, 0.2)
        return x


class ShapeDecoder(nn.Module):
    """Trunk of 5 layers with a DeepSDF-style skip into layer 5, then three 3-layer heads:
    position residual (conv6/7/8), scale (conv6N/7N/8N, sigmoid), colour (conv6SH/7SH/8SH,
    sigmoid). BatchNorm (batch statistics) + softplus after every hidden layer."""

    HEADS = ("", "N", "SH")

    def __init__(self, in_size, hsize=256, actv_fn="softplus"):
        super().__init__()
        self.hsize = hsize
        self.in_size = in_size
        h = hsize
        dims = {"conv1": (in_size, h), "conv2": (h, h), "conv3": (h, h), "conv4": (h, h),
                "conv5": (h + in_size, h)}
        out_dims = {"": 3, "SH": 3, "N": 1}
        for tag in ("", "SH", "N"):           # registration order of the reference
            dims[f"conv6{tag}"] = (h, h)
            dims[f"conv7{tag}"] = (h, h)
            dims[f"conv8{tag}"] = (h, out_dims[tag])
        for name in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv8",
                     "conv6SH", "conv7SH", "conv8SH", "conv6N", "conv7N", "conv8N"):
            ci, co = dims[name]
            setattr(self, name, nn.Conv1d(ci, co, 1))
        for name in ("bn1", "bn2", "bn3", "bn4", "bn5", "bn6", "bn7", "bn6N", "bn7N", "bn6SH", "bn7SH"):
            setattr(self, name, nn.BatchNorm1d(h))
        self.use_relu = actv_fn == "relu"

    def _layer(self, x, conv: str, bn: str):
        # on a HIP device: vendor GEMM forward + MFMA weight-gradient kernel, and one fused
        # BatchNorm(batch statistics)+softplus kernel pair (gaussianavatar_amd/fused.py);
        # on the CPU the same calls reduce to F.linear / BatchNorm1d / softplus
        c = getattr(self, conv)
        y = fused.linear(x, c.weight.squeeze(-1), c.bias)
        return fused.batchnorm_act(y, getattr(self, bn), "relu" if self.use_relu else "softplus")

    def _out(self, x, conv: str):
        c = getattr(self, conv)
        return fused.linear(x, c.weight.squeeze(-1), c.bias)

    def forward_points(self, x, raw_heads: bool = False, m_global=None):
        """x [M, in_size] -> (residual [M,3], scale [M,1], colour [M,3]); with raw_heads the scale and
        colour heads are returned as logits (the caller applies the sigmoids, fused.decode_pack).
        m_global: rows of the global batch when x holds one rank's share (BatchNorm statistics are then
        synchronised over the ranks; fused path only)."""
        act = (lambda t: t) if raw_heads else torch.sigmoid
        if fused.decoder_supported(self, x):
            # whole decoder on the fused MFMA layer kernels (activation-on-load, statistics in the
            # epilogue); the per-layer formulation below is the CPU / unsupported-shape path
            r, s, c = fused.decoder_mlp(self, x, m_global)
            return r, act(s), act(c)
        if m_global is not None and int(m_global) != x.shape[0] and self.training:
            # (evaluation mode normalises with the running statistics: nothing to synchronise, the per-layer
            # formulation below serves a rank's share of the rows as it is)
            raise NotImplementedError("synchronised BatchNorm statistics need the fused decoder path "
                                      "(HIP device, hsize 128, softplus, training mode)")
        x1 = self._layer(x, "conv1", "bn1")
        x2 = self._layer(x1, "conv2", "bn2")
        x3 = self._layer(x2, "conv3", "bn3")
        x4 = self._layer(x3, "conv4", "bn4")
        x5 = self._layer(torch.cat([x, x4], dim=1), "conv5", "bn5")
        outs = []
        for tag in self.HEADS:
            h6 = self._layer(x5, f"conv6{tag}", f"bn6{tag}")
            h7 = self._layer(h6, f"conv7{tag}", f"bn7{tag}")
            outs.append(self._out(h7, f"conv8{tag}"))
        return outs[0], act(outs[1]), act(outs[2])

    def forward(self, x):
        """Reference layout: x [B, C, L] -> ([B,3,L], [B,1,L], [B,3,L])."""
        B, C, L = x.shape
        r, s, c = self.forward_points(x.transpose(1, 2).reshape(B * L, C))
        back = lambda t: t.reshape(B, L, -1).transpose(1, 2)
        return back(r), back(s), back(c)


def conv4s2_gemm(x, weight):
    """Conv2d(k = 4, stride 2, padding 1, no bias) as im2col + ONE GEMM: [co, ci*16] x [B, ci*16, L]. On a HIP device the
    pose-encoder UNet's convolutions take this form instead of the vendor convolution library: MIOpen picks its solver from
    a per-user find database that does not exist in a fresh process — there (a cold box, e.g. the driver's) it falls back to
    its naive kernels (`naive_conv_ab_nonpacked_*`: 5.2 ms per backward call, 55 % of a stage-2 iteration, 90 instead of 129
    it/s; rocprofv3 of a cold process: profiles/r04_stage2_cold_miopen_kernel_stats.txt). The maps are tiny (64^2 .. 4^2), the
    GEMMs come precompiled with rocBLAS; autograd differentiates unfold / matmul / fold."""
    B, ci, H, W = x.shape
    co = weight.shape[0]
    cols = F.unfold(x, kernel_size=4, padding=1, stride=2)                 # [B, ci*16, (H/2)(W/2)]
    return torch.matmul(weight.reshape(co, ci * 16), cols).reshape(B, co, H // 2, W // 2)


def convT4s2_gemm(x, weight, bias=None):
    """ConvTranspose2d(k = 4, stride 2, padding 1): ONE GEMM [co*16, ci] x [B, ci, L] + col2im (see conv4s2_gemm)."""
    B, ci, H, W = x.shape
    co = weight.shape[1]                                                    # weight [ci, co, 4, 4]
    cols = torch.matmul(weight.reshape(ci, co * 16).t(), x.reshape(B, ci, H * W))
    y = F.fold(cols, output_size=(2 * H, 2 * W), kernel_size=4, padding=1, stride=2)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


class _Down(nn.Module):
    """LeakyReLU(0.2) -> conv4x4/s2 -> BN(affine=False) (modules.py:62-78)."""

    def __init__(self, ci, co, use_bn=True, use_relu=True):
        super().__init__()
        self.use_bn, self.use_relu = use_bn, use_relu
        self.conv = nn.Conv2d(ci, co, kernel_size=4, stride=2, padding=1, bias=False)
        if use_bn:
            self.bn = nn.BatchNorm2d(co, affine=False)

    def forward(self, x):
        x = conv4s2_gemm(x, self.conv.weight) if x.is_cuda else self.conv(x)
        return self.bn(x) if self.use_bn else x


class _Up(nn.Module):
    """ReLU -> convT4x4/s2 -> BN(affine=False) -> cat skip (modules.py:81-111)."""

    def __init__(self, ci, co, use_bn=True, use_bias=False, up_mode="upconv", use_dropout=False):
        super().__init__()
        self.use_bn, self.use_dropout = use_bn, use_dropout
        if up_mode == "upconv":
            self.up = nn.ConvTranspose2d(ci, co, kernel_size=4, stride=2, padding=1, bias=use_bias)
        else:
            self.up = nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2, align_corners=False),
                                    nn.Conv2d(ci, co, kernel_size=3, padding=1, stride=1))
        if use_bn:
            self.bn = nn.BatchNorm2d(co, affine=False)
        if use_dropout:
            self.drop = nn.Dropout(0.5)

    def forward(self, x, skip=None):
        x = F.relu(x)
        if x.is_cuda and isinstance(self.up, nn.ConvTranspose2d):
            x = convT4s2_gemm(x, self.up.weight, self.up.bias)
        else:
            x = self.up(x)
        if self.use_bn:
            x = self.bn(x)
        if self.use_dropout:
            x = self.drop(x)
        return x if skip is None else torch.cat([x, skip], 1)


class UnetNoCond5DS(nn.Module):
    """5-down / 5-up UNet. NOTE the reference's Conv2DBlock applies LeakyReLU *in place* on its
    input (modules.py:70), so every skip tensor d1..d4 that is later concatenated has already
    been LeakyReLU'd by the next down block — reproduced explicitly here."""

    def __init__(self, input_nc=3, output_nc=3, nf=64, up_mode="upconv", use_dropout=False,
                 return_lowres=False, return_2branches=False):
        super().__init__()
        assert not return_2branches, "two-branch variant is unused by the reference's hot path"
        self.conv1 = _Down(input_nc, nf, use_bn=False, use_relu=False)
        self.conv2 = _Down(nf, 2 * nf)
        self.conv3 = _Down(2 * nf, 4 * nf)
        self.conv4 = _Down(4 * nf, 8 * nf)
        self.conv5 = _Down(8 * nf, 8 * nf, use_bn=False)
        self.upconv1 = _Up(8 * nf, 8 * nf, up_mode=up_mode)
        self.upconv2 = _Up(16 * nf, 4 * nf, up_mode=up_mode, use_dropout=use_dropout)
        self.upconv3 = _Up(8 * nf, 2 * nf, up_mode=up_mode, use_dropout=use_dropout)
        self.upconv4 = _Up(4 * nf, nf, up_mode=up_mode)
        self.upconv5 = _Up(2 * nf, output_nc, use_bn=False, use_bias=True, up_mode=up_mode)

    def forward(self, x):
        if fused.unet_supported(self, x):
            # HIP device, the reference's configuration: the whole encoder as one native call each way on hand-written
            # implicit-GEMM kernels (csrc/ganet_unet.hip; channels-last, activations and BatchNorm applied on load)
            return fused.unet_forward(self, x)
        a1 = F.leaky_relu(self.conv1(x), 0.2)          # = d1 after the in-place activation
        a2 = F.leaky_relu(self.conv2(a1), 0.2)
        a3 = F.leaky_relu(self.conv3(a2), 0.2)
        a4 = F.leaky_relu(self.conv4(a3), 0.2)
        d5 = self.conv5(a4)
        u1 = self.upconv1(d5, a4)
        u2 = self.upconv2(u1, a3)
        u3 = self.upconv3(u2, a2)
        u4 = self.upconv4(u3, a1)
        return self.upconv5(u4)


class POP_no_unet(nn.Module):
    def __init__(self, c_geom=64, geom_layer_type="conv", nf=64, hsize=256, up_mode="upconv",
                 use_dropout=False, uv_feat_dim=2):
        super().__init__()
        self.geom_layer_type = geom_layer_type
        if geom_layer_type == "conv":
            self.geom_proc_layers = GeomConvLayers(c_geom, c_geom, c_geom, use_relu=False)
        elif geom_layer_type == "unet":
            self.geom_proc_layers = UnetNoCond5DS(c_geom, c_geom, nf, up_mode, use_dropout)
        elif geom_layer_type is not None:
            raise NotImplementedError(f"geom_layer_type={geom_layer_type!r} (the reference default is 'conv')")
        self.decoder = ShapeDecoder(in_size=uv_feat_dim + c_geom, hsize=hsize, actv_fn="softplus")

    @staticmethod
    def _is_broadcast(t: torch.Tensor) -> bool:
        return t.shape[0] > 1 and t.stride(0) == 0

    def _separable_bilinear(self, uv_loc, feat_res: int, uv_res: int):
        """If uv_loc[b, i*S+j] = (u_i, v_j) for every b (true for getIdxMap_torch's grid,
        /root/reference/utils/general_utils.py:165-176), return the two [S, R] matrices of
        bilinear weights that reproduce F.grid_sample(align_corners=False, zero padding) at those
        locations; otherwise None. The (one-off, synchronising) check is cached per uv tensor."""
        key = (uv_loc.data_ptr(), uv_loc._version, tuple(uv_loc.shape), tuple(uv_loc.stride()), feat_res)
        cache = getattr(self, "_bilinear_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        S, R = uv_res, feat_res
        result = None
        if uv_loc.shape[1] == S * S and uv_loc.shape[2] == 2:
            g = uv_loc.detach().reshape(uv_loc.shape[0], S, S, 2)
            u, v = g[0, :, 0, 0], g[0, 0, :, 1]
            same = bool(((g[..., 0] == u[None, :, None]) & (g[..., 1] == v[None, None, :])).all())
            if same:
                def weights(coord):
                    # grid_sample: x = 2*coord - 1 ; ix = ((x + 1) * R - 1) / 2 ; taps floor(ix), +1
                    ix = (((coord * 2 - 1.0) + 1.0) * R - 1.0) / 2.0
                    i0 = torch.floor(ix)
                    w1 = ix - i0
                    i0 = i0.long()
                    Wm = torch.zeros(S, R + 2, dtype=coord.dtype, device=coord.device)
                    rows = torch.arange(S, device=coord.device)
                    Wm[rows, (i0 + 1).clamp(0, R + 1)] += (1.0 - w1) * ((i0 >= -1) & (i0 <= R)).to(coord.dtype)
                    Wm[rows, (i0 + 2).clamp(0, R + 1)] += w1 * ((i0 + 1 >= -1) & (i0 + 1 <= R)).to(coord.dtype)
                    return Wm[:, 1:R + 1].contiguous()      # columns -1 and R are the zero padding
                # uv_to_grid transposes: output (i, j) samples input row <- v_i, column <- u_j
                result = (weights(v), weights(u))
        self._bilinear_cache = (key, result)
        return result

    def _bilinear_taps(self, mats):
        """Tap lists (fused.bilinear_taps) of the two weight matrices, cached with them."""
        cache = getattr(self, "_taps_cache", None)
        if cache is None or cache[0] is not mats:
            self._taps_cache = (mats, (fused.bilinear_taps(mats[0]), fused.bilinear_taps(mats[1])))
        return self._taps_cache[1]

    def forward_points(self, pose_featmap, geom_featmap, uv_loc, dedup: bool = True, raw_heads: bool = False,
                       rows=None, m_global=None):
        """-> (residuals [B,HW,3], scales [B,HW,1], colours [B,HW,3]); raw_heads: scale/colour logits.
        rows = (r0, r1): evaluate the decoder on texels r0..r1 only (outputs [b, r1-r0, .]; data-parallel
        texel sharding, batch-invariant inputs only); m_global: decoder rows of the global batch when this
        rank evaluates a share of them (BatchNorm statistics synchronised over ranks).

        If `dedup` and the inputs are batch-invariant (stage 1: pose_featmap None, geom_featmap
        and uv_loc expanded views of single maps) the net runs once and the result is expanded."""
        B = geom_featmap.shape[0]
        shared = (dedup and pose_featmap is None and self._is_broadcast(geom_featmap)
                  and (uv_loc.shape[0] == 1 or self._is_broadcast(uv_loc)))
        if shared:
            # [0].unsqueeze(0) rather than [:1]: a size-1 batch dimension that keeps the expand's
            # stride 0 makes MIOpen treat the tensor as non-packed and run its naive conv kernels
            # (14-30 ms per conv instead of 50 us)
            geom_featmap, uv_loc = geom_featmap[0].unsqueeze(0), uv_loc[0].unsqueeze(0)
        if self.geom_layer_type is not None:
            geom_featmap = self.geom_proc_layers(geom_featmap)
        if pose_featmap is not None:
            ready = getattr(pose_featmap, "_ga_ready", None)      # produced on a side stream (AvatarModel._pose_features)
            if ready is not None:
                torch.cuda.current_stream(pose_featmap.device).wait_event(ready)
        pix = geom_featmap if pose_featmap is None else pose_featmap + geom_featmap
        feat_res = geom_featmap.shape[2]
        uv_res = int(uv_loc.shape[1] ** 0.5)
        b, C = pix.shape[0], pix.shape[1]
        HW = uv_loc.shape[1]
        mats = self._separable_bilinear(uv_loc, feat_res, uv_res) if (pix.is_cuda and feat_res != uv_res) else None
        if uv_loc.shape[0] != b:          # one uv map for the whole batch (it is batch-invariant)
            uv_loc = uv_loc.expand(b, -1, -1)
        if feat_res != uv_res:
            pad = fused.decoder_input_pad(self.decoder, pix.new_empty(1)) if mats is not None else 0
            if mats is not None and pad and C == 64 and uv_loc.shape[-1] == 2:
                # separable query grid + fused decoder: one kernel writes the decoder's input rows
                # (bilinear 2x2 taps, uv columns, zero padding) — no dense GEMMs, no cat
                taps = self._bilinear_taps(mats)
                x = fused.upsample_cat(pix, uv_loc, taps[0], taps[1], C + 2 + pad)
                if rows is not None:
                    assert b == 1, "texel sharding applies to the batch-invariant (stage-1) decoder"
                    x = x[rows[0]:rows[1]]
                    HW = rows[1] - rows[0]
                r, s, c = self.decoder.forward_points(x, raw_heads=raw_heads, m_global=m_global)
                r, s, c = (t.reshape(b, HW, -1) for t in (r, s, c))
                if shared:
                    r, s, c = (t.expand(B, -1, -1) for t in (r, s, c))
                return r, s, c
            if mats is not None:
                # the query grid is separable (the reference's texel-centre grid): bilinear
                # up-sampling = two small dense GEMMs that write the point-major layout directly
                # (no scatter-add backward, no transposes) — same weights as grid_sample
                Wr, Wc = mats
                featP = pix.permute(0, 2, 3, 1)                               # [b, R, R, C]
                t1 = torch.matmul(Wc, featP)                                  # [b, R(p), S(j), C]
                pts = torch.matmul(Wr, t1.reshape(b, feat_res, uv_res * C))   # [b, S(i), S(j)*C]
                pts = pts.reshape(b, HW, C)
            else:
                pix = F.grid_sample(pix, uv_to_grid(uv_loc, uv_res), mode="bilinear", align_corners=False)
                pts = pix.reshape(b, C, HW).transpose(1, 2)
        else:
            pts = pix.reshape(b, C, HW).transpose(1, 2)
        parts = [pts, uv_loc]
        pad = fused.decoder_input_pad(self.decoder, pts)
        if pad:       # the fused decoder wants 8-float k-blocks: append the zero columns here, for free
            zkey = (b, HW, pad, pts.device)
            if getattr(self, "_zero_pad", (None, None))[0] != zkey:
                self._zero_pad = (zkey, pts.new_zeros(b, HW, pad))
            parts.append(self._zero_pad[1])
        x = torch.cat(parts, dim=2)                                           # [b, HW, C+2 (+pad)]
        if rows is not None:
            raise NotImplementedError("texel sharding needs the fused up-sampling path (c_geom 64, separable uv grid)")
        r, s, c = self.decoder.forward_points(x.reshape(b * HW, x.shape[2]), raw_heads=raw_heads, m_global=m_global)
        r, s, c = (t.reshape(b, HW, -1) for t in (r, s, c))
        if shared:
            r, s, c = (t.expand(B, -1, -1) for t in (r, s, c))
        return r, s, c

    def forward(self, pose_featmap, geom_featmap, uv_loc):
        """Reference signature/layout (network.py:39-83): -> ([B,3,HW], [B,1,HW], [B,3,HW])."""
        r, s, c = self.forward_points(pose_featmap, geom_featmap, uv_loc, dedup=False)
        return r.transpose(1, 2), s.transpose(1, 2), c.transpose(1, 2)
