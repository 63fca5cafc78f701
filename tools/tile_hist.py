"""Histogram of per-tile list lengths on the bench scene (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from gaussianavatar_amd import rasterizer
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
torch.manual_seed(0)
mp, npar, op = default_params(batch_size=2, num_points=200_000, image_width=1024, image_height=1024, num_frames=16)
m = AvatarModel(mp, npar, op, train=True); m.training_setup()
for s in range(4):
    batch = collate_frames([m.train_dataset[(2 * s + k) % 16] for k in range(2)], "cuda")
    with torch.no_grad():
        image, *_ = m.train_stage1(batch, 7)
    torch.cuda.synchronize()
    print("step", s, "status", rasterizer.last_status())
