"""Diagnostic: cost of the first `run_steps` calls of a fresh engine at different lengths, per engine call."""
import sys, time
sys.path.insert(0, ".")
from dist_classicrl_amd import _lib
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ExponentialSchedule
algo = OptimalQLearningBase(1_000_000, 16, 0.99, seed=0)
env = HashTabularEnv(128, 1_000_000, 16, seed=1)
rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995))
lib = _lib.load()
T = {}
class W:
    def __init__(s, f, k): s.f, s.k = f, k
    def __call__(s, *a):
        t = time.perf_counter(); r = s.f(*a); T[s.k] = T.get(s.k, 0.0) + time.perf_counter() - t; return r
for name in ("qe_schedule_plan", "qe_rollout_begin", "qe_rollout_end", "qe_episode_log", "qe_env_observe", "qe_env_restore"):
    setattr(lib, name, W(getattr(lib, name), name[3:]))
sd = None
for steps in [int(a) for a in sys.argv[1:]]:
    T.clear()
    t0 = time.perf_counter()
    _, _, _, sd = rt.run_steps(steps, env, sd)
    el = time.perf_counter() - t0
    print(f"run_steps({steps}): {el * 1e3:.2f} ms (device {rt.last_stats['kernel_ms']:.2f} ms); " + ", ".join(f"{k} {v * 1e3:.2f}" for k, v in T.items()))
