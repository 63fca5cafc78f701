"""cProfile of the SYNC=1 enqueue loop (tools/cpu_enqueue.py): which Python functions hold the host time of an iteration."""
import cProfile, pstats, sys, os, io
os.environ["SYNC"] = "1"
sys.path.insert(0, os.getcwd())
import runpy
pr = cProfile.Profile()
ns = runpy.run_path("tools/cpu_enqueue.py", run_name="__main__")
step = ns["step"]
import torch
for i in range(20): step(i)
torch.cuda.synchronize()
pr.enable()
for i in range(200): step(i)
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
