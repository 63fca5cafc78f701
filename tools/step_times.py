"""Per-step wall time (synchronised) of the first N train iterations in a fresh process (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
from gaussianavatar_amd.losses import l1_loss_w, ssim

torch.manual_seed(0)
B = 2
STAGE = int(os.environ.get("STAGE", 1))      # STAGE=2: the stage-2 iteration (pose-encoder UNet + per-frame decoder)
mp, npar, op = default_params(batch_size=B, num_points=200_000, image_width=1024, image_height=1024, num_frames=16, train_stage=STAGE)
model = AvatarModel(mp, npar, op, train=True)
model.training_setup()
if STAGE == 2:
    with torch.no_grad():
        model.net.decoder.conv8N.weight.mul_(0.01); model.net.decoder.conv8N.bias.fill_(-5.65)
ds = model.train_dataset
dev = torch.device("cuda")
batches = [collate_frames([ds[(s * B + k) % len(ds)] for k in range(B)], dev) for s in range(8)]
gt = torch.ones(B, 3, 1024, 1024, device=dev)
import gc
if os.environ.get('GA_GC_FREEZE'):
    gc.collect(); gc.freeze()
ts = []
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    t0 = sync()
    if STAGE == 2:
        image, points, pose_loss, offset_loss = model.train_stage2(batches[i % 8], 7)
    else:
        image, points, offset_loss, geo_loss, scale_loss = model.train_stage1(batches[i % 8], 7)
    t1 = sync()
    if STAGE == 2:
        loss = 0.8 * l1_loss_w(image, gt) + 0.2 * (1 - ssim(image, gt)) + offset_loss + 10 * pose_loss
    else:
        loss = 0.8 * l1_loss_w(image, gt) + 0.2 * (1 - ssim(image, gt)) + offset_loss + geo_loss + scale_loss
    t2 = sync()
    model.zero_grad(1); loss.backward()
    t3 = sync()
    model.step(1)
    t4 = sync()
    ts.append(1e3 * (t4 - t0))
    if i > 2 and ts[-1] > 15:
        print("SPIKE_WINDOW", " ".join(f"{name}={getattr(time, 'clock_gettime_ns')(getattr(time, name))}" for name in
              ("CLOCK_MONOTONIC", "CLOCK_MONOTONIC_RAW", "CLOCK_BOOTTIME", "CLOCK_REALTIME")), f"dur_ns={int((t3-t2)*1e9)}", flush=True)
        print(f"slow step {i}: fwd {1e3*(t1-t0):.1f} loss {1e3*(t2-t1):.1f} bwd {1e3*(t3-t2):.1f} opt {1e3*(t4-t3):.1f}  "
              f"gc {gc.get_count()} mem {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
print(" ".join(f"{t:.1f}" for t in ts))
print("reserved GiB", torch.cuda.memory_reserved() / 2**30, "allocs", torch.cuda.memory_stats()["num_device_alloc"])
