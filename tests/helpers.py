"""Shared helpers for the parity tests (oracle side only; no product imports here)."""

from __future__ import annotations

from pathlib import Path

import numpy as np

from oracle.envs import GridLakeEnv, HashTabularEnv, RiggedBanditVecEnv, TicTacToeVecEnv
from oracle.qlearn_oracle import OracleQLearning, OracleRuntime, OracleSchedule

GOLDEN = Path(__file__).resolve().parent / "golden"

# must mirror tests/golden/make_golden.py:TRACE_CASES (name -> env spec, steps, dtype, schedules, learn)
TRACE_CASES = {
    "c1_grid_n1": (("grid", 1, 10), 80, "f8", "bench", "iter"),
    "grid4_n1": (("grid", 1, 4), 300, "f8", "bench", "iter"),
    "grid4_n16": (("grid", 16, 4), 100, "f4", "const", "iter"),
    "grid4_n16_f8": (("grid", 16, 4), 100, "f8", "const", "iter"),
    "c2_hash_n128": (("hash", 128, 10000, 8, False), 50, "f4", "bench", "iter"),
    "c2_hash_n128_const": (("hash", 128, 10000, 8, False), 50, "f4", "const", "iter"),
    "hash_dense_n256": (("hash", 256, 64, 16, False), 40, "f4", "const", "iter"),
    "hash_dense_n256_f8": (("hash", 256, 64, 16, False), 40, "f8", "const", "iter"),
    "hash_dense_n256_vec": (("hash", 256, 64, 16, False), 40, "f8", "const", "vec"),
    "c5_hash_masked_n128": (("hash", 128, 500, 64, True), 40, "f4", "const", "iter"),
    "hash_masked_a9_n64": (("hash", 64, 300, 9, True), 40, "f8", "const", "iter"),
    "bandit_n4": (("bandit", 4, 5), 23, "f8", "kat", "iter"),
    "bandit_n128": (("bandit", 128, 7), 30, "f4", "const", "iter"),
    "ttt_n64": (("ttt", 64), 60, "f8", "bench", "iter"),
    "ttt_n128_f4": (("ttt", 128), 50, "f4", "const", "iter"),
}


def make_oracle_env(spec):
    if spec[0] == "hash":
        _, n, S, A, masked = spec
        return HashTabularEnv(n, S, A, seed=1, masked=masked)
    if spec[0] == "grid":
        return GridLakeEnv(spec[1], side=spec[2], seed=1)
    if spec[0] == "ttt":
        return TicTacToeVecEnv(spec[1], seed=1)
    return RiggedBanditVecEnv(spec[1], episode_len=spec[2])


def schedule_params(kind):
    """(lr, eps) as (kind, value, min, decay) tuples."""
    if kind == "bench":
        return ("exponential", 0.1, 1e-5, 0.995), ("exponential", 1.0, 0.01, 0.995)
    if kind == "const":
        return ("constant", 0.1, None, None), ("constant", 0.1, None, None)
    return ("constant", 1.0, None, None), ("linear", 0.05, None, 0.001)


def run_oracle_trace(spec, steps, dt, sched, learn_mode, gamma=0.99, seed=0):
    env = make_oracle_env(spec)
    algo = OracleQLearning(env.state_size, env.action_size, gamma, seed=seed, dtype=np.dtype(dt))
    lr_p, eps_p = schedule_params(sched)
    rt = OracleRuntime(algo, OracleSchedule(*lr_p), OracleSchedule(*eps_p), learn_mode=learn_mode)
    rt.trace = []
    states, _ = env.reset()
    acc = np.zeros(env.num_agents, dtype=np.float32)
    history = []
    for _ in range(steps):
        states, _ = rt.run_single_step(env, states, acc, history)
    obs = states["observation"] if isinstance(states, dict) else states
    return {
        "actions": np.stack([a for a, _, _ in rt.trace]),
        "eps": np.array([e for _, e, _ in rt.trace]),
        "lr": np.array([v for _, _, v in rt.trace]),
        "q": algo.q_table,
        "history": np.array(history, dtype=np.float32),
        "final_obs": np.asarray(obs, dtype=np.int32),
        "agent_rewards": acc,
        "final_sched": np.array([rt.lr_schedule.get_value(), rt.exploration_rate_schedule.get_value()]),
    }


def dense_from_sparse(idx, val, shape, dtype):
    q = np.zeros(shape, dtype=dtype)
    q.ravel()[idx] = val
    return q
