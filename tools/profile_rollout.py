"""Diagnostic: cProfile of tools/quick_rollout.py's run() (host side of one rollout configuration)."""
import cProfile, io, pstats, sys
sys.path.insert(0, "tools")
import quick_rollout

args = sys.argv[1].split(",")
pr = cProfile.Profile()
pr.enable()
quick_rollout.run(int(args[0]), int(args[1]), int(args[2]), None if args[3] == "sched" else float(args[3]), int(args[4]))
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue())
