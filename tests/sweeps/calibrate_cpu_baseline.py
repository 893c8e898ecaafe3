"""Build-container only: time the REAL reference loop next to the oracle's NumPy restatement on the
same environment objects and injected draws, to show that the oracle's single-core speed is a fair
stand-in for the reference's single_thread runtime.

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python tests/sweeps/calibrate_cpu_baseline.py
"""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from dist_classicrl.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl.algorithms.runtime.base_runtime import BaseRuntime
from dist_classicrl.schedules.exponential_schedule import ExponentialSchedule
from oracle.draws import InjectedDraws
from oracle.envs import HashTabularEnv
from oracle.qlearn_oracle import OracleQLearning, OracleRuntime, OracleSchedule


class Harness(BaseRuntime):
    step_counter = 0
    def init_training(self): pass
    def close_training(self): pass
    def run_steps(self, *a, **k): raise NotImplementedError
    def _choose_actions(self, states):
        n = len(states["observation"]) if isinstance(states, dict) else len(states)
        self.algorithm._rng.begin(self.step_counter, n, self.exploration_rate_schedule.get_value())
        self.step_counter += 1
        return super()._choose_actions(states)


def bench(make, n, steps=400, warm=50):
    rt, env = make()
    states, _ = env.reset()
    acc, hist = np.zeros(n, dtype=np.float32), []
    for _ in range(warm):
        states, _ = rt.run_single_step(env, states, acc, hist)
    t0 = time.perf_counter()
    for _ in range(steps):
        states, _ = rt.run_single_step(env, states, acc, hist)
    return steps * n / (time.perf_counter() - t0)


try:
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[-1]})  # one core, like single_thread mode
except (AttributeError, OSError):
    pass
ROUNDS = 9
results = {}
for n, S, A, masked in [(128, 1_000_000, 16, False), (128, 10_000, 8, False), (1024, 1_000_000, 64, True)]:
    def ref():
        algo = OptimalQLearningBase(S, A, 0.99, seed=0)
        algo.q_table.fill(0.0)
        algo._rng = algo._np_rng = InjectedDraws(0)
        return Harness(algo, ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995)), \
            HashTabularEnv(n, S, A, seed=1, masked=masked)
    def ora():
        algo = OracleQLearning(S, A, 0.99, seed=0)
        algo.q_table.fill(0.0)
        return OracleRuntime(algo, OracleSchedule("exponential", 0.1, 1e-5, 0.995),
                             OracleSchedule("exponential", 1.0, 0.01, 0.995)), \
            HashTabularEnv(n, S, A, seed=1, masked=masked)
    steps = 400 if n <= 128 else 60
    rs, os_ = [], []
    for _ in range(ROUNDS):  # interleaved rounds: machine noise hits both sides alike
        rs.append(bench(ref, n, steps))
        os_.append(bench(ora, n, steps))
    r, o = np.median(rs), np.median(os_)
    print(f"n={n} S={S} A={A} masked={masked}: reference {r:,.0f} env-steps/s, oracle {o:,.0f} env-steps/s, ratio {o / r:.3f}")
    results[f"n{n}_S{S}_A{A}{'_masked' if masked else ''}"] = {
        "reference_env_steps_per_s": [float(x) for x in rs], "oracle_env_steps_per_s": [float(x) for x in os_],
        "median_ratio_oracle_over_reference": float(o / r),
        "per_round_ratio": [float(b / a) for a, b in zip(rs, os_)]}
out = Path(__file__).resolve().parents[2] / "profiles" / "r02_cpu_calibration.json"
json.dump({"tool": "tests/sweeps/calibrate_cpu_baseline.py (build container, one pinned core, interleaved rounds)",
           "rounds": ROUNDS, "results": results}, open(out, "w"), indent=1)
print("written", out)
