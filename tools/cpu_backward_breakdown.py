"""Where the host time of loss.backward() goes: wall time inside each custom Function's backward (dev tool)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
from gaussianavatar_amd.losses import l1_loss_w, ssim, weighted_sum
import gaussianavatar_amd.fused as fused, gaussianavatar_amd.rasterizer as rasterizer, gaussianavatar_amd.lbs as lbs, gaussianavatar_amd.losses as losses

acc = collections.defaultdict(float)
def wrap(cls):
    for name in ("forward", "backward"):
        f = getattr(cls, name)
        def g(*a, _f=f, _k=cls.__name__ + "." + name, **kw):
            t = time.perf_counter()
            try:
                return _f(*a, **kw)
            finally:
                acc[_k] += time.perf_counter() - t
        setattr(cls, name, staticmethod(g))
for mod in (fused, rasterizer, lbs, losses):
    for k, v in list(vars(mod).items()):
        if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
            wrap(v)

torch.manual_seed(0)
B = 2
mp, npar, op = default_params(batch_size=B, num_points=200_000, image_width=1024, image_height=1024, num_frames=16)
model = AvatarModel(mp, npar, op, train=True)
model.training_setup()
ds = model.train_dataset
dev = torch.device("cuda")
batches = [collate_frames([ds[(s * B + k) % len(ds)] for k in range(B)], dev) for s in range(8)]
gt = torch.ones(B, 3, 1024, 1024, device=dev)
tb = [0.0, 0.0, 0.0]
def step(i, rec):
    t0 = time.perf_counter()
    image, points, offset_loss, geo_loss, scale_loss = model.train_stage1(batches[i % 8], 7)
    l = op.lambda_dssim
    loss = weighted_sum([scale_loss, offset_loss, l1_loss_w(image, gt), ssim(image, gt), geo_loss],
                        [op.lambda_scale, op.lambda_rgl, 1.0 - l, -l, 1.0], bias=l)
    t1 = time.perf_counter()
    model.zero_grad(1)
    loss.backward()
    t2 = time.perf_counter()
    model.step(1)
    t3 = time.perf_counter()
    if rec:
        tb[0] += t1 - t0; tb[1] += t2 - t1; tb[2] += t3 - t2
for i in range(30):
    step(i, False)
torch.cuda.synchronize()
acc.clear()
N = 50
t0 = time.perf_counter()
for i in range(N):
    step(i, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("GA_DEV", os.environ.get("GA_DEV", ""), " wall/iter %.3f ms  CPU loop %.3f ms  drain %.1f ms" % (1e3 * (t2 - t0) / N, 1e3 * (t1 - t0) / N, 1e3 * (t2 - t1)))
print("  forward+loss %.3f  backward %.3f  optimizer %.3f ms" % tuple(1e3 * v / N for v in tb))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-40s %.3f ms/iter" % (k, 1e3 * v / N))
print("  device allocs", torch.cuda.memory_stats()["num_device_alloc"], "reserved GiB %.2f" % (torch.cuda.memory_reserved() / 2**30))
