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
    if kind == "nan":  # diverging: lr = 1 with colliding learn_vec increments overflows float32 within tens of steps
        return ("constant", 1.0, None, None), ("constant", 0.3, None, None)
    if kind == "explore":  # every pick exploratory: no greedy selection ever meets a NaN row
        return ("constant", 0.25, None, None), ("constant", 1.0, None, None)
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


def run_oracle_chunks(spec, chunks, dt, sched, learn_mode, gamma=0.99, seed=0, q0=None):
    """The closed loop in `chunks` consecutive run_steps-sized pieces.  Returns one dict per chunk with the state at
    its end (q, history so far, obs, acc); a chunk in which the reference's selection raises IndexError (a NaN row
    maximum under a NumPy variant: random.choice([]), q_learning_optimal.py:470, :563) ends the list with
    {"raised": True}."""
    env = make_oracle_env(spec)
    algo = OracleQLearning(env.state_size, env.action_size, gamma, seed=seed, dtype=np.dtype(dt))
    if q0 is not None:
        algo.q_table = np.array(q0, dtype=np.dtype(dt))
    lr_p, eps_p = schedule_params(sched)
    rt = OracleRuntime(algo, OracleSchedule(*lr_p), OracleSchedule(*eps_p), learn_mode=learn_mode)
    rt.trace = []
    states, _ = env.reset()
    acc = np.zeros(env.num_agents, dtype=np.float32)
    history, out = [], []
    for k in chunks:
        try:
            with np.errstate(all="ignore"):
                for _ in range(k):
                    states, _ = rt.run_single_step(env, states, acc, history)
        except IndexError:
            out.append({"raised": True})
            break
        obs = states["observation"] if isinstance(states, dict) else states
        out.append({"raised": False, "q": algo.q_table.copy(), "history": np.array(history, dtype=np.float32),
                    "final_obs": np.asarray(obs, dtype=np.int32).copy(), "agent_rewards": acc.copy(),
                    "actions": np.stack([a for a, _, _ in rt.trace]) if rt.trace else None})
    return out


def nan_table(S, A, dt, cells, seed):
    """The initial table of the NaN-regime goldens (same construction as tests/golden/make_golden_r3.py)."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((S, A)).astype(dt)
    if cells:
        q.ravel()[rng.choice(S * A, size=cells, replace=False)] = np.nan
    return q


def spec_shape(spec):
    if spec[0] == "ttt":
        return 19683, 9
    if spec[0] == "bandit":
        return 1, 2
    if spec[0] == "grid":
        return spec[2] * spec[2], 4
    return spec[2], spec[3]


def golden_nan_trace(g, name, spec, dt, cells, tseed):
    """Chunk-end states of one closed-loop NaN-regime golden (real reference) in run_oracle_chunks' format."""
    S, A = spec_shape(spec)
    base = nan_table(S, A, dt, cells, tseed) if cells else np.zeros((S, A), dtype=dt)
    u = np.uint32 if np.dtype(dt).itemsize == 4 else np.uint64
    n_ok, raised = int(g[f"{name}/n_ok"][0]), int(g[f"{name}/raised_in_chunk"][0])
    actions = g[f"{name}/actions"].astype(np.int32)
    out = []
    for c in range(n_ok):
        q = (g[f"{name}/qx{c}"] ^ base.view(u)).view(np.dtype(dt))
        assert np.array_equal(np.flatnonzero(np.isnan(q)), g[f"{name}/nan_cells{c}"])
        out.append({"raised": False, "q": q, "history": g[f"{name}/history{c}"], "final_obs": g[f"{name}/final_obs{c}"],
                    "agent_rewards": g[f"{name}/agent_rewards{c}"], "actions": actions})
    if raised >= 0:
        out.append({"raised": True})
    return out, base
