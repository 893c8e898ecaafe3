"""Diagnostic: cProfile of the host side of bench.py (run under torch.distributed.run for the sync path)."""
import cProfile, pstats, sys, io
sys.argv = ["bench.py"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    exec(compile(open("bench.py").read(), "bench.py", "exec"), {"__name__": "__main__", "__file__": "bench.py"})
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats("gpu_rollout|delta_sync|schedules|_lib|distributed_c10d|ctypes|numpy|Work|method", 40)
    sys.stderr.write(s.getvalue())
