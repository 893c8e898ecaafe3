"""Generate the golden vectors under ``tests/golden/`` by running the REAL reference.

Run in the build container only (the reference cannot travel to the GPU box):

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does: imports the reference's ``OptimalQLearningBase``, ``BaseRuntime`` and schedules,
plants :class:`oracle.draws.InjectedDraws` as ``algo._rng`` / ``algo._np_rng`` (the technique of the
reference's own runtime tests, ``tests/dist_classicrl/algorithms/runtime/
test_q_learning_runtimes.py:17-45``), drives it with this build's integer environments
(``oracle/envs.py``) and stores inputs + outputs as small ``.npz`` files.  The files hold data only.
``single_thread_runtime.py`` cannot be imported here (it needs gymnasium at module level), so its
five-line ``run_steps`` loop is driven directly on the imported ``BaseRuntime.run_single_step``.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from dist_classicrl.algorithms.base_algorithms.q_learning_optimal import (  # noqa: E402
    OptimalQLearningBase,
)
from dist_classicrl.algorithms.runtime.base_runtime import BaseRuntime  # noqa: E402
from dist_classicrl.schedules.constant_schedule import ConstantSchedule  # noqa: E402
from dist_classicrl.schedules.exponential_schedule import ExponentialSchedule  # noqa: E402
from dist_classicrl.schedules.linear_schedule import LinearSchedule  # noqa: E402

from oracle.draws import InjectedDraws  # noqa: E402
from oracle.envs import GridLakeEnv, HashTabularEnv, RiggedBanditVecEnv, TicTacToeVecEnv  # noqa: E402

OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))
from make_golden_cases import LEARN_CASES, SELECT_CASES, TRACE_CASES  # noqa: E402


def _tie_rich_table(rng, S, A, dtype):
    """Q-values from a tiny value set so that arg-max ties are common; some all-zero rows."""
    q = rng.choice(np.array([0.0, 0.25, 0.5, 1.0, -0.5]), size=(S, A)).astype(dtype)
    q[rng.random(S) < 0.2] = 0
    return q


def _masks(rng, n, A):
    m = (rng.random((n, A)) < 0.5).astype(np.int32)
    m[np.arange(n), rng.integers(A, size=n)] = 1  # at least one valid action per agent
    return m


# ----------------------------------------------------------------------------- selection


def gen_select():
    out = {}
    rng = np.random.default_rng(1234)
    for k, (method, S, A, n, masked, eps, det, dt) in enumerate(SELECT_CASES):
        seed, step = 100 + k, 7 * k + (1 << 33) * (k % 2)
        algo = OptimalQLearningBase(S, A, 0.9, seed=0)
        algo.q_table = _tie_rich_table(rng, S, A, np.dtype(dt))
        shim = InjectedDraws(seed)
        algo._rng = algo._np_rng = shim
        states = rng.integers(S, size=n).astype(np.int32)
        masks = _masks(rng, n, A) if masked else None
        shim.begin(step, n, eps, deterministic=det)
        fn = getattr(algo, method)
        if method == "choose_actions_vec":
            acts = fn(states, eps, deterministic=det)
        elif method == "choose_masked_actions_vec":
            acts = fn(states, masks, eps, deterministic=det)
        else:
            acts = fn(states, eps, deterministic=det, action_masks=masks)
        p = f"c{k}_"
        out[p + "meta"] = np.array([S, A, n, int(masked), int(det), seed], dtype=np.int64)
        out[p + "step"] = np.array([step], dtype=np.uint64)
        out[p + "eps"] = np.array([eps])
        out[p + "q"] = algo.q_table
        out[p + "states"] = states
        if masked:
            out[p + "masks"] = masks.astype(np.int8)
        out[p + "actions"] = np.asarray(acts, dtype=np.int32)
    out["n_cases"] = np.array([len(SELECT_CASES)])
    np.savez_compressed(OUT / "select.npz", **out)
    print("select.npz:", len(SELECT_CASES), "cases")


# ----------------------------------------------------------------------------- learning


def gen_learn():
    out = {}
    rng = np.random.default_rng(4321)
    for k, (S, A, n, masked, dt, lr, gamma) in enumerate(LEARN_CASES):
        q0 = (rng.standard_normal((S, A)) * 2).astype(dt)
        states = rng.integers(S, size=n).astype(np.int32)
        actions = rng.integers(A, size=n).astype(np.int32)
        rewards = rng.random(n).astype(np.float32)
        next_states = rng.integers(S, size=n).astype(np.int32)
        terminated = rng.random(n) < 0.2
        masks = _masks(rng, n, A) if masked else None
        p = f"c{k}_"
        out[p + "meta"] = np.array([S, A, n, int(masked)], dtype=np.int64)
        out[p + "hyper"] = np.array([lr, gamma])
        out[p + "q0"] = q0
        out[p + "states"], out[p + "actions"], out[p + "rewards"] = states, actions, rewards
        out[p + "next_states"], out[p + "terminated"] = next_states, terminated
        if masked:
            out[p + "masks"] = masks.astype(np.int8)
        for name in ("learn", "learn_vec"):
            algo = OptimalQLearningBase(S, A, gamma, seed=0)
            algo.q_table = q0.copy()
            getattr(algo, name)(states, actions, rewards, next_states, terminated, lr, masks)
            assert algo.q_table.dtype == q0.dtype
            out[p + "q_" + name] = algo.q_table
    out["n_cases"] = np.array([len(LEARN_CASES)])
    np.savez_compressed(OUT / "learn.npz", **out)
    print("learn.npz:", len(LEARN_CASES), "cases")


# ----------------------------------------------------------------------------- closed loop
class _Harness(BaseRuntime):
    """The reference ``BaseRuntime`` with the abstract methods filled in as no-ops; the only added
    behaviour is telling the draw shim which vector step a selection belongs to and logging."""

    step_counter = 0
    log = None

    def init_training(self):
        pass

    def close_training(self):
        pass

    def run_steps(self, steps, env, curr_state_dict=None):
        raise NotImplementedError

    def _choose_actions(self, states):
        n = len(states["observation"]) if isinstance(states, dict) else len(states)
        eps = self.exploration_rate_schedule.get_value()
        self.algorithm._rng.begin(self.step_counter, n, eps)
        actions = super()._choose_actions(states)
        self.log.append((np.asarray(actions, dtype=np.int32).copy(), eps, self.lr_schedule.get_value()))
        self.step_counter += 1
        return actions


def _make_env(spec):
    kind = spec[0]
    if kind == "hash":
        _, n, S, A, masked = spec
        return HashTabularEnv(n, S, A, seed=1, masked=masked)
    if kind == "grid":
        return GridLakeEnv(spec[1], side=spec[2], seed=1)
    if kind == "ttt":
        return TicTacToeVecEnv(spec[1], seed=1)
    return RiggedBanditVecEnv(spec[1], episode_len=spec[2])




def _schedules(kind):
    if kind == "bench":  # benchmarks/throughput_benchmark.py:53-59,157-166
        return ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995)
    if kind == "const":
        return ConstantSchedule(0.1), ConstantSchedule(0.1)
    if kind == "nan":  # diverging (tests/helpers.py:schedule_params)
        return ConstantSchedule(1.0), ConstantSchedule(0.3)
    if kind == "explore":
        return ConstantSchedule(0.25), ConstantSchedule(1.0)
    return ConstantSchedule(1.0), LinearSchedule(0.05, 0.001)  # "kat"


def gen_traces():
    out = {}
    names = []
    for name, spec, steps, dt, sched, learn_fn in TRACE_CASES:
        env = _make_env(spec)
        algo = OptimalQLearningBase(env.state_size, env.action_size, 0.99, seed=0)
        algo.q_table = algo.q_table.astype(dt)
        algo._rng = algo._np_rng = InjectedDraws(0)
        if learn_fn == "learn_vec":
            algo.learn = algo.learn_vec  # what the reference's commented-out branch (:923-933) does
        lr, eps = _schedules(sched)
        rt = _Harness(algo, lr, eps)
        rt.log = []
        states, _infos = env.reset()
        n = env.num_agents
        agent_rewards = np.zeros(n, dtype=np.float32)
        history = []
        for _ in range(steps):  # single_thread_runtime.py:63-64
            states, _infos = rt.run_single_step(env, states, agent_rewards, history)
        assert algo.q_table.dtype == np.dtype(dt)
        obs = states["observation"] if isinstance(states, dict) else states
        nz = np.flatnonzero(algo.q_table)
        p = name + "/"
        out[p + "actions"] = np.stack([a for a, _, _ in rt.log])
        out[p + "eps"] = np.array([e for _, e, _ in rt.log])
        out[p + "lr"] = np.array([v for _, _, v in rt.log])
        out[p + "q_idx"] = nz.astype(np.int64)
        out[p + "q_val"] = algo.q_table.ravel()[nz]
        out[p + "history"] = np.array(history, dtype=np.float32)
        out[p + "final_obs"] = np.asarray(obs, dtype=np.int32)
        out[p + "agent_rewards"] = agent_rewards
        out[p + "final_sched"] = np.array([lr.get_value(), eps.get_value()])
        names.append(name)
        print(f"  trace {name}: {len(history)} episodes, {nz.size} non-zero cells")
    out["names"] = np.array(names)
    np.savez_compressed(OUT / "traces.npz", **out)
    print("traces.npz:", len(names), "traces")


if __name__ == "__main__":
    gen_select()
    gen_learn()
    gen_traces()
