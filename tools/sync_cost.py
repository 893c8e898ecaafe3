"""Diagnostic: what the closing synchronize of a timed region costs -- 0.3 us on an idle stream, 9-11 us right behind a rollout
whose results the host already holds (the kernel publishes them itself: the runtime sees the end of the launch that much
later; spinning on hipStreamQuery instead of the blocking wait does not shorten it)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from dist_classicrl_amd import _lib
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ExponentialSchedule
algo = OptimalQLearningBase(1_000_000, 16, 0.99, seed=0)
algo.set_engine_option(_lib.OPT_EVENT_TIMING, 0)
env = HashTabularEnv(128, 1_000_000, 16, seed=1)
rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995))
_, _, _, sd = rt.run_steps(200, env, None)
lib = algo._lib
# idle-stream synchronize
t0 = time.perf_counter()
for _ in range(2000): lib.qe_synchronize(algo.handle)
print("qe_synchronize on an idle stream: %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
# right after a 20-step call
tot = 0.0; tc = 0.0
for _ in range(300):
    t0 = time.perf_counter()
    _, _, _, sd = rt.run_steps(20, env, sd)
    t1 = time.perf_counter()
    lib.qe_synchronize(algo.handle)
    t2 = time.perf_counter()
    tc += t1 - t0; tot += t2 - t1
print("20-step call %.1f us, synchronize right behind it %.2f us" % (tc / 300 * 1e6, tot / 300 * 1e6))
