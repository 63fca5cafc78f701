"""CPU enqueue time vs GPU time of a train iteration (dev tool): is the loop launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
from gaussianavatar_amd.losses import l1_loss_w, ssim, weighted_sum

torch.manual_seed(0)
B = 2
mp, npar, op = default_params(batch_size=B, num_points=200_000, image_width=1024, image_height=1024, num_frames=16)
model = AvatarModel(mp, npar, op, train=True)
model.training_setup()
ds = model.train_dataset
dev = torch.device("cuda")
batches = [collate_frames([ds[(s * B + k) % len(ds)] for k in range(B)], dev) for s in range(8)]
gt = torch.ones(B, 3, 1024, 1024, device=dev)


SYNC = os.environ.get("SYNC", "0") == "1"      # drain the GPU before every iteration: pure enqueue time per section,
                                                # i.e. what a loop with a per-iteration .item() (the reference's train.py:101) pays


def step(i, marks=None):
    if SYNC:
        torch.cuda.synchronize()
    t = [time.perf_counter()]
    image, points, offset_loss, geo_loss, scale_loss = model.train_stage1(batches[i % 8], 7)
    t.append(time.perf_counter())
    l = op.lambda_dssim
    loss = weighted_sum([scale_loss, offset_loss, l1_loss_w(image, gt), ssim(image, gt), geo_loss],
                        [op.lambda_scale, op.lambda_rgl, 1.0 - l, -l, 1.0], bias=l)
    t.append(time.perf_counter())
    model.zero_grad(1)
    loss.backward()
    t.append(time.perf_counter())
    model.step(1)
    t.append(time.perf_counter())
    if SYNC:
        torch.cuda.synchronize()
        t.append(time.perf_counter())
    if marks is not None:
        marks.append([1e3 * (b - a) for a, b in zip(t, t[1:])])


for i in range(30):
    step(i)
torch.cuda.synchronize()
N = 60
marks = []
t0 = time.perf_counter()
for i in range(N):
    step(i, marks)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
import statistics
names = ["forward", "loss", "backward", "optimizer"] + (["drain (GPU tail after the last enqueue)"] if SYNC else [])
print("wall per iteration %.3f ms; CPU in the loop %.3f ms; final drain %.3f ms" % (1e3 * (t2 - t0) / N, 1e3 * (t1 - t0) / N, 1e3 * (t2 - t1)))
for j, n in enumerate(names):
    print("  CPU %-10s median %.3f ms" % (n, statistics.median(m[j] for m in marks)))
