"""Diagnostic: fixed cost of one persistent-kernel launch (device time of very short rollouts)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ConstantSchedule
algo = OptimalQLearningBase(1_000_000, 16, 0.99, seed=0)
env = HashTabularEnv(128, 1_000_000, 16, seed=1)
rt = GpuRolloutQLearning(algo, ConstantSchedule(0.1), ConstantSchedule(0.05))
try:
    _, _, _, sd = rt.run_steps(3000, env, None)
except ZeroDivisionError:
    sd = None
for steps in (1, 2, 10, 50, 100, 200, 1000):
    ms = []
    for rep in range(20):
        try:
            _, _, _, sd = rt.run_steps(steps, env, sd)
        except ZeroDivisionError:
            pass
        ms.append(rt.last_stats["kernel_ms"])
    print(steps, "steps:", round(float(np.median(ms)) * 1e3, 1), "us per launch")
