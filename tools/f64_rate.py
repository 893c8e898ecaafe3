"""Diagnostic: headline shape with a float64 table (the reference's default dtype)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ExponentialSchedule
for dt in (np.float32, np.float64):
    algo = OptimalQLearningBase(1_000_000, 16, 0.99, seed=0, dtype=dt)
    env = HashTabularEnv(128, 1_000_000, 16, seed=1)
    rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995))
    _, _, _, sd = rt.run_steps(2000, env, None)
    t0 = time.perf_counter()
    _, _, _, sd = rt.run_steps(20000, env, sd)
    el = time.perf_counter() - t0
    print(np.dtype(dt).name, f"{20000 * 128 / el / 1e6:.1f} M env-steps/s, device {rt.last_stats['kernel_ms'] / 20000 * 1e3:.2f} us/step")
