#!/usr/bin/env python
"""What the forward pass records for the backward pass on the headline frame (200k avatar Gaussians, 1024^2, the
bench's scale warm-up): (tile, Gaussian) pairs D, recorded segments, their entries = the (render block, Gaussian)
survivors = the 64-byte atomic gradient records one backward pass issues; per-block segment counts (the serial depth
of a forward wave). Run with GA_DEV=lib_dir=<variant> to compare block sizes (tools/build_gsr_variant.sh -DGSR_SUB=4)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd import _native, rasterizer
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
from gaussianavatar_amd.lbs import skin

torch.manual_seed(0)
N = int(os.environ.get("POINTS", 200000)); size = int(os.environ.get("SIZE", 1024)); it = int(os.environ.get("ITER", 7))
mp, npar, op = default_params(batch_size=1, num_points=N, image_width=size, image_height=size)
m = AvatarModel(mp, npar, op, train=True)
batch = collate_frames([m.train_dataset[0]], "cuda")
with torch.no_grad():
    live = m._body(m.pose.weight[:1], m.transl.weight[:1], None)
    _off, _scl, point_res, scales, colors = m._decode(1, None, it, True)
    pts = skin(m.query_points[:1], point_res, m.query_lbs[0], live.cano2live)[0].contiguous()
rs = rasterizer.GaussianRasterizationSettings(
    size, size, math.tan(float(batch["FovX"][0]) * 0.5), math.tan(float(batch["FovY"][0]) * 0.5), m.background, 1.0,
    batch["world_view_transform"][0], batch["full_proj_transform"][0], 0, batch["camera_center"][0], False, False)
color, radii, v, status = rasterizer.rasterize_with_state(rs, pts, colors[0].contiguous(), m.fix_opacity,
                                                          scales[0].contiguous(), m.fix_rotation)
D = status[0]
sc = v["seg_count"].long()
nseg = int(sc.sum())
edge = _native.gsr().gsr_render_block_edge()
T = sc.shape[0]
off = v["tile_offset"].long()
# entries per recorded segment: walk the slots of every block
Bk = sc.shape[1]
start = off[:-1]; end = off[1:]
first = Bk * ((start >> 6) + torch.arange(T, device=start.device))
cap = (end >> 6) - (start >> 6) + 1
info = v["seg_info"]
entries = 0
occupied = (end > start).nonzero().flatten().tolist()
for t in occupied:
    for b in range(Bk):
        c = int(sc[t, b])
        if c:
            s0 = int(first[t]) + b * int(cap[t])
            entries += int(info[s0:s0 + c, 1].long().sum())
print(f"block edge {edge}: P={N} D={D} pairs/frame, occupied tiles {len(occupied)}, segments {nseg}, "
      f"survivor records {entries} ({entries / max(D, 1):.2f} per pair, {entries / N:.2f} per Gaussian), "
      f"mean fill {entries / max(nseg, 1):.1f}/64; segments per block: mean {sc[sc > 0].float().mean():.1f} "
      f"max {int(sc.max())}, blocks with segments {int((sc > 0).sum())}")
