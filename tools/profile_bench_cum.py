"""Diagnostic: cProfile of bench.py, top functions by cumulative time."""
import cProfile, pstats, sys, io
sys.argv = ["bench.py"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    exec(compile(open("bench.py").read(), "bench.py", "exec"), {"__name__": "__main__", "__file__": "bench.py"})
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("dist_classicrl_amd|bench.py|numpy|ctypes", 30)
    sys.stderr.write(s.getvalue())
