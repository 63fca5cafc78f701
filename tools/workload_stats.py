#!/usr/bin/env python
"""Shape of the headline rasterizer workload (CPU, oracle): per-tile list lengths, contributors."""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussianavatar_amd.synthetic import make_assets, make_frames
from oracle import lbs_oracle as O
from oracle.gsr_oracle import RasterOracle

N, S, size = 200000, 512, 1024
a = make_assets(N, S)
f = make_frames(a, 2, size, size)
valid = a["valid_idx"].reshape(-1)
pts = a["query_posmap"].reshape(-1, 3)[valid]
w = a["lbs_map"].reshape(-1, 24)[valid]
A = O.joint_transforms(f["pose"][:1], f["transl"][:1], a["joints_rest"], torch.tensor(a["parents"]).long())
M = A @ torch.linalg.inv(a["cano_joint_mat"])
full = O.skin(pts[None], torch.zeros(1, N, 3), w[None], M)[0].numpy()
cam = f["camera"]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0035
R = RasterOracle()
rot = np.zeros((N, 4), np.float32); rot[:, 0] = 1
t = time.time()
st = R.forward(full, np.full((N, 3), 0.5, np.float32), np.ones(N, np.float32), np.full((N, 3), scale, np.float32), rot,
               viewmatrix=cam["world_view_transform"], projmatrix=cam["full_proj_transform"], bg=np.ones(3, np.float32),
               W=size, H=size, tanfovx=math.tan(cam["FovX"] / 2), tanfovy=math.tan(cam["FovY"] / 2))
cnt = (st["ranges"][:, 1] - st["ranges"][:, 0]).astype(np.int64)
nz = cnt[cnt > 0]
print("oracle fwd s", round(time.time() - t, 2), "D", st["D"], "visible", int((st["radii"] > 0).sum()), "radius mean/max", st["radii"][st["radii"] > 0].mean(), st["radii"].max())
print("tiles nonempty", len(nz), "of", len(cnt), "mean", nz.mean(), "median", np.median(nz), "p90", np.percentile(nz, 90), "max", nz.max())
print("hist", np.histogram(nz, bins=[1, 64, 256, 512, 1024, 2048, 4096, 8192, 1 << 20])[0])
nc = st["n_contrib"].reshape(size, size)
print("pixels with contrib", int((nc > 0).sum()), "mean n_contrib (those)", nc[nc > 0].mean(), "max", nc.max())
# per-tile max n_contrib vs list length
gx = size // 16
tmax = nc.reshape(gx, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
print("sum over tiles of max n_contrib", int(tmax.sum()), "vs sum n", int(cnt.sum()))
print("covered fraction of image", float((st["final_T"] < 0.5).mean()))
