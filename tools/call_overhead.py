"""Diagnostic: where the fixed cost of a short `run_steps` call goes (host side)."""
import cProfile, io, pstats, sys, time
sys.path.insert(0, ".")
import numpy as np
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ExponentialSchedule
algo = OptimalQLearningBase(1_000_000, 16, 0.99, seed=0)
env = HashTabularEnv(128, 1_000_000, 16, seed=1)
rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995))
_, _, _, sd = rt.run_steps(2000, env, None)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(300):
    _, _, _, sd = rt.run_steps(steps, env, sd)
print(f"{(time.perf_counter() - t0) / 300 * 1e6:.1f} us per run_steps({steps}) call")
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    _, _, _, sd = rt.run_steps(steps, env, sd)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue())
