"""Diagnostic: device time per step of the headline shape as a function of the launch length (no replica exchange)."""
import sys
sys.path.insert(0, ".")
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ExponentialSchedule
for chunk in (2000, 500, 100, 50):
    algo = OptimalQLearningBase(1_000_000, 16, 0.99, seed=0)
    env = HashTabularEnv(128, 1_000_000, 16, seed=1)
    rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995))
    rt._PIPELINE_CHUNK = chunk
    _, _, _, sd = rt.run_steps(2000, env, None)
    _, _, _, sd = rt.run_steps(20000, env, sd)
    st = rt.last_stats
    print(f"chunk {chunk}: device {st['kernel_ms'] / 20000 * 1e3:.3f} us/step over {st['launches']} launches")
