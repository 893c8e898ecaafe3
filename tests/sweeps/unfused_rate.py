"""Diagnostic: PCIe-inclusive rate of the UNFUSED API (choose_actions + host env.step + learn per vector step)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from oracle.envs import HashTabularEnv  # a host-side environment stands in for "arbitrary host env"
n, S, A = 128, 1_000_000, 16
algo = OptimalQLearningBase(S, A, 0.99, seed=0)
env = HashTabularEnv(n, S, A, seed=1)
states, _ = env.reset()
for phase, steps in (("warm", 200), ("timed", 2000)):
    t_sel = t_env = t_learn = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter(); actions = algo.choose_actions(states, 0.05); b = time.perf_counter()
        nxt, r, term, trunc, _ = env.step(actions); c = time.perf_counter()
        algo.learn(states, actions, r, nxt, term, 0.1); d = time.perf_counter()
        t_sel += b - a; t_env += c - b; t_learn += d - c
        states = nxt
    el = time.perf_counter() - t0
print(f"unfused: {steps * n / el / 1e6:.3f} M env-steps/s; per vector step: choose_actions {t_sel/steps*1e6:.0f} us, "
      f"host env {t_env/steps*1e6:.0f} us, learn {t_learn/steps*1e6:.0f} us")
print(f"engine calls only (choose_actions + learn): {steps * n / (t_sel + t_learn) / 1e6:.3f} M env-steps/s")
