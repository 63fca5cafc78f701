"""development: host time of every custom autograd backward of a stage-1 iteration (they run on autograd's device thread,
where cProfile does not look): the Function classes' backward methods wrapped with perf_counter."""
import os, sys, time, collections
sys.path.insert(0, os.getcwd())
os.environ["SYNC"] = "1"
import torch
from gaussianavatar_amd import fused, lbs, rasterizer, parallel, losses
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls):
    real = cls.backward
    def timed(ctx, *a):
        t = time.perf_counter()
        try:
            return real(ctx, *a)
        finally:
            e = acc[cls.__name__]; e[0] += time.perf_counter() - t; e[1] += 1
    cls.backward = staticmethod(timed)
for mod in (fused, lbs, rasterizer, parallel, losses):
    for v in list(vars(mod).values()):
        if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
            wrap(v)
import runpy
ns = runpy.run_path("tools/cpu_enqueue.py", run_name="__main__")
acc.clear()
step = ns["step"]
t0 = time.perf_counter()
for i in range(200): step(i)
print("per iteration, ms:")
tot = 0.0
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:28s} {1e3 * s / 200:.3f}  ({n // 200} calls)")
    tot += s
print("  sum", round(1e3 * tot / 200, 3))
