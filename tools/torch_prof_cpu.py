"""torch.profiler CPU view of the training iteration: where the host time goes (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
from gaussianavatar_amd.losses import l1_loss_w, ssim, weighted_sum
torch.manual_seed(0)
B = 2
mp, npar, op = default_params(batch_size=B, num_points=200_000, image_width=1024, image_height=1024, num_frames=16)
m = AvatarModel(mp, npar, op, train=True); m.training_setup()
batches = [collate_frames([m.train_dataset[(2 * s + k) % 16] for k in range(B)], "cuda") for s in range(4)]
gt = torch.ones(B, 3, 1024, 1024, device="cuda")
def step(i):
    image, points, offset_loss, geo_loss, scale_loss = m.train_stage1(batches[i % 4], 7)
    l = op.lambda_dssim
    loss = weighted_sum([scale_loss, offset_loss, l1_loss_w(image, gt), ssim(image, gt), geo_loss],
                        [op.lambda_scale, op.lambda_rgl, 1.0 - l, -l, 1.0], bias=l)
    m.zero_grad(1); loss.backward(); m.step(1)
for i in range(10): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(10): step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=50))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=25, max_name_column_width=50))
