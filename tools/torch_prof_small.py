"""Which aten ops launch the fill/copy/elementwise kernels of one iteration (dev tool):
kernel events grouped by (kernel name, launching aten op, python source line)."""
import collections, os, sys
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
    loss = weighted_sum([scale_loss, offset_loss, l1_loss_w(image, gt), ssim(image, gt), geo_loss], [0.03, 10.0, 0.8, -0.2, 1.0], bias=0.2)
    m.zero_grad(1); loss.backward(); m.step(1)
for i in range(6): step(i)
torch.cuda.synchronize()
NIT = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(NIT): step(i)
    torch.cuda.synchronize()
ev = prof.events()
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
agg = collections.defaultdict(lambda: [0, 0.0])
for e in cpu:
    for k in e.kernels:
        nm = k.name[:60]
        small = any(s in nm for s in ("fill", "Fill", "copy", "Copy", "elementwise", "reduce_kernel", "CatArray", "index", "multi_tensor"))
        if not small:
            continue
        where = str(e.input_shapes)[:90]
        key = (nm, e.name[:28], where)
        agg[key][0] += 1
        agg[key][1] += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("small-kernel time per iteration: %.1f us" % (tot / NIT))
for (nm, op_, where), (n, us) in rows[:60]:
    print("%6.1f us  %4.1f/it  %-44s %-28s %s" % (us / NIT, n / NIT, nm[:44], op_, where))
