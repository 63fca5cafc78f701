"""cProfile of tools/reference_loop.py (where the host time of the reference's own loop goes); train.py silences
sys.stdout (--quiet), so the table goes to sys.__stdout__."""
import cProfile, pstats, sys, os, io
sys.argv = ["tools/reference_loop.py", "--iters", "256", "--frames", "32"]
sys.path.insert(0, os.getcwd())
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("tools/reference_loop.py", run_name="__main__")
finally:
    pr.disable()
    s = io.StringIO()
    ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
    ps.print_stats(70)
    sys.__stdout__.write(s.getvalue()[:16000])
    sys.__stdout__.flush()
