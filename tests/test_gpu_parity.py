"""GPU parity tests: the HIP engine, called through the C ABI (ctypes), against

* the golden vectors produced by the real reference (bit-exact actions AND bit-exact Q-values in
  both table dtypes -- tighter than the 1e-6 the north star asks for);
* the oracle on seeded inputs at sizes it finishes in seconds;
* size-independent properties at BASELINE.json's full sizes.

Nothing here reads /root/reference (it does not exist on the GPU box).
"""

import os

import numpy as np
import pytest

from helpers import GOLDEN, TRACE_CASES, dense_from_sparse, run_oracle_trace, schedule_params
from golden.make_golden_cases import LEARN_CASES, SCALE_TRACE_CASES, SELECT_CASES

pytestmark = pytest.mark.gpu


def _product():
    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd import environments, schedules

    return OptimalQLearningBase, GpuRolloutQLearning, environments, schedules


def make_device_env(spec):
    _, _, envs, _ = _product()
    if spec[0] == "hash":
        _, n, S, A, masked = spec
        return envs.HashTabularEnv(n, S, A, seed=1, masked=masked)
    if spec[0] == "grid":
        return envs.GridLakeEnv(spec[1], side=spec[2], seed=1)
    if spec[0] == "ttt":
        return envs.TicTacToeEnv(spec[1], seed=1)
    return envs.RiggedTwoArmedBanditVecEnv(spec[1], episode_len=spec[2])


def make_schedule(p):
    _, _, _, sch = _product()
    kind, value, lo, decay = p
    if kind == "exponential":
        return sch.ExponentialSchedule(value, lo, decay)
    if kind == "linear":
        return sch.LinearSchedule(value, decay)
    return sch.ConstantSchedule(value)


# ------------------------------------------------------------------------------- selection
@pytest.mark.parametrize("k", range(len(SELECT_CASES)))
def test_select_matches_reference_golden(k):
    Algo = _product()[0]
    g = np.load(GOLDEN / "select.npz")
    method = SELECT_CASES[k][0]
    S, A, n, masked, det, seed = (int(v) for v in g[f"c{k}_meta"])
    eps, step = float(g[f"c{k}_eps"][0]), int(g[f"c{k}_step"][0])
    q = g[f"c{k}_q"]
    algo = Algo(S, A, 0.9, seed=seed, dtype=q.dtype)
    algo.q_table = q
    algo.step_counter = step
    masks = g[f"c{k}_masks"] if masked else None
    if method == "choose_actions_vec":
        acts = algo.choose_actions_vec(g[f"c{k}_states"], eps, deterministic=bool(det))
    elif method == "choose_masked_actions_vec":
        acts = algo.choose_masked_actions_vec(g[f"c{k}_states"], masks, eps, deterministic=bool(det))
    else:
        acts = getattr(algo, method)(g[f"c{k}_states"], eps, deterministic=bool(det), action_masks=masks)
    assert acts.dtype == np.int32
    assert np.array_equal(acts, g[f"c{k}_actions"])  # bit-exact action indices
    assert algo.step_counter == step + 1


@pytest.mark.parametrize("n", [1, 3, 40, 300])
@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("eps", [0.0, 0.5, 1.0])
def test_agents_without_valid_action_follow_each_variant(n, det, eps):
    """Agent 0's mask is all zeros.  The reference's list variants return -1 for it (:302, :348); its
    NumPy variants tie every action at -inf, so the greedy pick is uniform over ALL actions and only an
    exploratory pick raises IndexError (:470, :618-628).  The oracle was checked against the real
    reference on exactly this grid (identical on all 96 combinations)."""
    from oracle.draws import InjectedDraws
    from oracle.qlearn_oracle import OracleQLearning

    Algo = _product()[0]
    S, A = 6, 5
    q = (np.arange(S * A, dtype=np.float64).reshape(S, A) % 7).astype(np.float32)
    states = (np.arange(n) % S).astype(np.int32)
    masks = np.ones((n, A), dtype=np.int32)
    masks[0] = 0
    for method in ("choose_actions", "choose_actions_iter", "choose_actions_vec_iter", "choose_masked_actions_vec"):
        results = []
        for side in ("oracle", "product"):
            if side == "oracle":
                algo = OracleQLearning(S, A, 0.9, dtype=np.float32)
                algo._rng = algo._np_rng = shim = InjectedDraws(7)
                shim.begin(3, n, eps, deterministic=det)
            else:
                algo = Algo(S, A, 0.9, seed=7)
                algo.step_counter = 3
            algo.q_table = q.copy()
            try:
                if method == "choose_masked_actions_vec":
                    out = getattr(algo, method)(states, masks, eps, deterministic=det)
                else:
                    out = getattr(algo, method)(states, eps, deterministic=det, action_masks=masks)
                results.append(("ok", np.asarray(out).tolist()))
            except IndexError:
                results.append(("IndexError", None))
        assert results[0] == results[1], (method, results)


def test_agents_without_valid_action_match_reference_golden():
    """The same situation against the fixture generated from the real reference (tests/golden/empty_mask.npz)."""
    Algo = _product()[0]
    g = np.load(GOLDEN / "empty_mask.npz")
    S, A, seed, step = (int(v) for v in g["meta"])
    for k in range(int(g["count"])):
        n, det, eps100 = (int(v) for v in g[f"c{k}_cfg"])
        method, eps = str(g[f"c{k}_method"]), eps100 / 100.0
        algo = Algo(S, A, 0.9, seed=seed, dtype=np.float64)
        algo.q_table = g["q"]
        algo.step_counter = step
        states = (np.arange(n) % S).astype(np.int32)
        masks = np.ones((n, A), dtype=np.int32)
        masks[0] = 0
        try:
            if method == "choose_masked_actions_vec":
                got = getattr(algo, method)(states, masks, eps, deterministic=bool(det))
            else:
                got = getattr(algo, method)(states, eps, deterministic=bool(det), action_masks=masks)
            raised = 0
        except IndexError:
            got, raised = np.zeros(0, dtype=np.int32), 1
        assert raised == int(g[f"c{k}_raised"]), (k, method, n, det, eps)
        assert np.array_equal(np.asarray(got, dtype=np.int32), g[f"c{k}_actions"]), (k, method, n, det, eps)


# ------------------------------------------------------------------------------- learning
@pytest.mark.parametrize("k", range(len(LEARN_CASES)))
@pytest.mark.parametrize("fn", ["learn", "learn_vec"])
def test_learn_matches_reference_golden(k, fn):
    Algo = _product()[0]
    g = np.load(GOLDEN / "learn.npz")
    S, A, n, masked = (int(v) for v in g[f"c{k}_meta"])
    lr, gamma = (float(v) for v in g[f"c{k}_hyper"])
    q0 = g[f"c{k}_q0"]
    algo = Algo(S, A, gamma, seed=0, dtype=q0.dtype)
    algo.q_table = q0
    masks = g[f"c{k}_masks"] if masked else None
    getattr(algo, fn)(g[f"c{k}_states"], g[f"c{k}_actions"], g[f"c{k}_rewards"], g[f"c{k}_next_states"],
                      g[f"c{k}_terminated"], lr, masks)
    got, want = np.asarray(algo.q_table), g[f"c{k}_q_{fn}"]
    assert got.dtype == want.dtype
    # sequential semantics (learn) and np.add.at accumulation order (learn_vec) are both reproduced
    # exactly: bit-exact Q-values in either dtype.
    assert np.array_equal(got, want)


@pytest.mark.parametrize(("S", "A", "n", "masked", "dt"), [
    (20, 300, 64, False, "f4"), (12, 1000, 40, True, "f8"), (7, 257, 90, True, "f4"),  # A > 256: wave-per-row kernels
    (3, 1, 50, False, "f4"),  # a single action
    (9, 5, 3000, False, "f4"), (9, 5, 3000, True, "f8"),  # far more transitions than cells
])
@pytest.mark.parametrize("fn", ["learn", "learn_vec"])
def test_learn_matches_oracle_on_extreme_shapes(S, A, n, masked, dt, fn):
    """Shapes outside the golden set (the reference has no table this wide), against the oracle that the
    golden vectors pin: bit-exact Q-values."""
    from oracle.qlearn_oracle import OracleQLearning

    Algo = _product()[0]
    rng = np.random.default_rng(S * 1000 + A)
    q0 = rng.standard_normal((S, A)).astype(dt)
    s, a = rng.integers(S, size=n).astype(np.int32), rng.integers(A, size=n).astype(np.int32)
    r, s2 = rng.random(n).astype(np.float32), rng.integers(S, size=n).astype(np.int32)
    term = rng.random(n) < 0.15
    masks = None
    if masked:
        masks = (rng.random((n, A)) < 0.5).astype(np.int8)
        masks[np.arange(n), rng.integers(A, size=n)] = 1  # at least one valid action per row
    algo = Algo(S, A, 0.9, seed=0, dtype=np.dtype(dt))
    algo.q_table = q0
    ref = OracleQLearning(S, A, 0.9, dtype=np.dtype(dt))
    ref.q_table = q0.copy()
    getattr(algo, fn)(s, a, r, s2, term, 0.05, masks)
    getattr(ref, fn)(s, a, r, s2, term, 0.05, masks)
    assert np.array_equal(np.asarray(algo.q_table), ref.q_table)


def test_empty_batches_are_no_ops():
    """n = 0 (the reference returns an empty int32 array / leaves the table alone)."""
    Algo = _product()[0]
    algo = Algo(10, 4, 0.9, seed=0)
    algo.q_table = np.arange(40, dtype=np.float32).reshape(10, 4)
    e = np.array([], dtype=np.int32)
    for det in (False, True):
        for masks in (None, np.zeros((0, 4), dtype=np.int32)):
            out = algo.choose_actions(e, 0.1, deterministic=det, action_masks=masks)
            assert out.dtype == np.int32 and out.shape == (0,)
    for fn in ("learn", "learn_vec", "learn_iter"):
        getattr(algo, fn)(e, e, np.array([], dtype=np.float32), e, np.array([], dtype=bool), 0.1)
    assert np.array_equal(np.asarray(algo.q_table), np.arange(40, dtype=np.float32).reshape(10, 4))


def test_learn_vec_many_collisions_stay_exact():
    """20 000 transitions on 800 cells: far more colliding increments than the LDS structures of the ordered
    path hold, so they are accumulated in index-ordered batches -- still exactly np.add.at's order and
    rounding (float64 add, rounded into the table dtype), hence bit-exact."""
    from oracle.qlearn_oracle import OracleQLearning

    Algo = _product()[0]
    rng = np.random.default_rng(5)
    S, A, n = 50, 16, 20000
    q0 = rng.standard_normal((S, A)).astype(np.float32)
    s, a = rng.integers(S, size=n).astype(np.int32), rng.integers(A, size=n).astype(np.int32)
    r, s2 = rng.random(n).astype(np.float32), rng.integers(S, size=n).astype(np.int32)
    term = rng.random(n) < 0.1
    algo = Algo(S, A, 0.9, seed=0)
    algo.q_table = q0
    algo.learn_vec(s, a, r, s2, term, 0.01)
    ref = OracleQLearning(S, A, 0.9, dtype=np.float32)
    ref.q_table = q0.copy()
    ref.learn_vec(s, a, r, s2, term, 0.01)
    assert np.array_equal(np.asarray(algo.q_table), ref.q_table)


# ------------------------------------------------------------------------------- closed loop
# (every run below collects an action trace, and a traced rollout of up to 512 agents runs the GENERIC build of the
# persistent kernel whatever QE_OPT_LANE_ORDERED_PATH says: the builds untraced rollouts get -- sparse, dataflow, full
# -- are compared with the oracle in tests/test_gpu_shipped_builds.py)
PATHS = ["stepwise", "persistent", "wide", "wide_listed", "turnstile", "turnstile_reread"]


def _run_product_trace(spec, steps, dt, sched, mode, gamma=0.99, seed=0, path="auto"):
    Algo, Runtime, _, _ = _product()
    env = make_device_env(spec)
    algo = Algo(env.state_size, env.action_size, gamma, seed=seed, dtype=np.dtype(dt))
    if path == "persistent" and (env.action_size > 64 or env.num_agents > 512):
        pytest.skip("more than 512 agents / 64 actions: the persistent kernel does not apply")
    if path == "turnstile_reread":  # the turnstile path without value forwarding in the progress words
        if mode != "iter" or dt != "f4":
            pytest.skip("value forwarding exists for float32 learn_iter rollouts only: nothing to switch off")
        from dist_classicrl_amd import _lib
        algo.set_engine_option(_lib.OPT_TURN_FORWARD, 0)
        path = "turnstile"
    if path == "wide_listed":  # the compacted-list rounds (automatic from 16384 agents), seven rounds
        from dist_classicrl_amd import _lib
        algo.set_rollout_path("wide")
        algo.set_engine_option(_lib.OPT_LISTED_MIN_AGENTS, 1)
        algo.set_engine_option(_lib.OPT_TOKEN_ROUNDS, 7)  # >= 6: lists are used
    else:
        algo.set_rollout_path(path)
    if os.environ.get("QE_TEST_STAMP_BITS"):  # tests/sweeps/fuzz_parity.py: hashed touch counters of a random size
        from dist_classicrl_amd import _lib
        algo.set_engine_option(_lib.OPT_STAMP_HASH_BITS, int(os.environ["QE_TEST_STAMP_BITS"]))
    lr_p, eps_p = schedule_params(sched)
    rt = Runtime(algo, make_schedule(lr_p), make_schedule(eps_p), learn_mode=mode)
    rt.trace_actions = True
    try:
        _avg, history, _env, sd = rt.run_steps(steps, env, None)
    except ZeroDivisionError:  # reference quirk when no episode ends; state is still valid
        history = []
        sd = None
    obs, acc = env.observe()
    return {
        "actions": rt.trace_actions,
        "q": np.asarray(algo.q_table),
        "history": np.array(history, dtype=np.float32),
        "final_obs": obs["observation"] if isinstance(obs, dict) else obs,
        "agent_rewards": acc,
        "final_sched": np.array([rt.lr_schedule.get_value(), rt.exploration_rate_schedule.get_value()]),
        "stats": rt.last_stats,
        "state_dict": sd,
    }


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", list(TRACE_CASES))
def test_rollout_matches_reference_golden(name, path):
    g = np.load(GOLDEN / "traces.npz")
    spec, steps, dt, sched, mode = TRACE_CASES[name]
    got = _run_product_trace(spec, steps, dt, sched, mode, path=path)
    assert np.array_equal(got["actions"], g[f"{name}/actions"])  # every action of every step
    want_q = dense_from_sparse(g[f"{name}/q_idx"], g[f"{name}/q_val"], got["q"].shape, got["q"].dtype)
    assert np.array_equal(got["q"], want_q)  # bit-exact Q-values (iter and vec alike)
    assert np.array_equal(got["history"], g[f"{name}/history"])
    assert np.array_equal(got["final_obs"], g[f"{name}/final_obs"])
    assert np.array_equal(got["agent_rewards"], g[f"{name}/agent_rewards"])
    assert np.array_equal(got["final_sched"], g[f"{name}/final_sched"])


@pytest.mark.parametrize("path", ["stepwise", "wide", "wide_listed", "turnstile", "turnstile_reread", "auto"])
@pytest.mark.parametrize("case", SCALE_TRACE_CASES, ids=[c[0] for c in SCALE_TRACE_CASES])
def test_rollout_matches_reference_golden_at_scale(case, path):
    """SURVEY 8(c)'s "shrunk C3" (4096 agents on the 1e6 x 16 table: the turnstile path's chains of row sharers) and
    "C5-small" (1024 masked agents, 64 actions), generated from the REAL reference (tests/golden/make_golden_r3.py)."""
    name, spec, steps, dt, sched, learn_fn = case
    g = np.load(GOLDEN / "traces_scale.npz")
    got = _run_product_trace(spec, steps, dt, sched, "iter" if learn_fn == "learn" else "vec", path=path)
    assert np.array_equal(got["actions"], g[f"{name}/actions"].astype(np.int32))
    idx = np.cumsum(g[f"{name}/q_idx_delta"].astype(np.int64))
    want_q = dense_from_sparse(idx, g[f"{name}/q_val"], got["q"].shape, got["q"].dtype)
    assert np.array_equal(got["q"], want_q)
    assert np.array_equal(got["history"], g[f"{name}/history"])
    assert np.array_equal(got["final_obs"], g[f"{name}/final_obs"])
    assert np.array_equal(got["agent_rewards"], g[f"{name}/agent_rewards"])
    assert np.array_equal(got["final_sched"], g[f"{name}/final_sched"])
    if path in ("turnstile", "auto"):
        from dist_classicrl_amd import _lib
        assert _lib.decode_variant(got["stats"]["kernel_variant"])["path"] == "turnstile"


@pytest.mark.parametrize(
    ("spec", "steps", "dt", "mode"),
    [
        (("hash", 1024, 5000, 16, False), 30, "f4", "iter"),
        (("hash", 1024, 5000, 16, False), 30, "f8", "iter"),
        (("hash", 4096, 300, 8, False), 12, "f4", "iter"),  # > 1024 involved agents: batched ordered path
        (("hash", 8192, 150, 4, False), 8, "f4", "iter"),  # eight batches, long same-cell chains
        (("hash", 3000, 40000, 16, False), 10, "f8", "iter"),  # batches whose rows overflow the LDS row cache
        (("hash", 5000, 600, 40, True), 6, "f4", "iter"),
        (("hash", 512, 2000, 64, True), 20, "f4", "iter"),
        (("hash", 300, 100000, 32, False), 25, "f4", "iter"),
        (("hash", 700, 1000, 4, False), 25, "f8", "iter"),
        (("hash", 1000, 200, 12, True), 10, "f4", "iter"),
        (("grid", 200, 6), 40, "f4", "iter"),
        (("bandit", 300, 4), 12, "f8", "iter"),
        (("hash", 1024, 5000, 16, False), 30, "f4", "vec"),
        (("hash", 600, 150, 8, False), 15, "f8", "vec"),
        (("hash", 500, 400, 64, True), 15, "f4", "vec"),
        (("hash", 256, 40, 16, False), 60, "f4", "iter"),  # persistent kernel, every step contested
        (("hash", 250, 90, 13, True), 50, "f8", "iter"),
        (("hash", 120, 3000, 32, True), 80, "f4", "vec"),
        (("hash", 6000, 100000, 12, True), 12, "f4", "vec"),  # > 2048 involved agents in learn_vec: batched, exact
        (("hash", 2500, 40, 16, False), 20, "f4", "vec"),
        (("hash", 4096, 2000, 9, True), 16, "f8", "vec"),
        (("hash", 128, 60, 16, False), 60, "f4", "iter"),  # two wavefronts of agents on 60 states: every step complex
        (("hash", 64, 25, 8, False), 50, "f4", "iter"),
        (("hash", 128, 4000, 16, False), 80, "f4", "iter"),  # mostly quiet / two-toucher steps, a few complex ones
        (("hash", 128, 700, 32, False), 60, "f4", "iter"),  # persistent, 8 lanes per row
        (("hash", 64, 300, 64, True), 60, "f8", "iter"),  # persistent, 16 lanes per row
        (("hash", 60, 500, 50, False), 70, "f4", "iter"),  # 16 lanes per row, A not a multiple of 4
        (("hash", 500, 100000, 8, False), 100, "f4", "iter"),
        (("grid", 1000, 5), 30, "f4", "iter"),
        (("grid", 500, 5), 40, "f8", "iter"),
        (("hash", 512, 700, 4, False), 40, "f4", "iter"),
        (("hash", 200, 50, 20, True), 40, "f4", "vec"),
        (("bandit", 128, 5), 25, "f4", "vec"),
        (("ttt", 256), 80, "f4", "iter"),
        (("ttt", 700), 40, "f8", "iter"),
        (("ttt", 100), 60, "f4", "vec"),
    ],
)
@pytest.mark.parametrize("path", PATHS)
def test_rollout_matches_oracle_seeded(spec, steps, dt, mode, path):
    if path == "persistent" and spec[1] > 512:
        pytest.skip("more than 512 agents: the persistent kernel does not apply")
    want = run_oracle_trace(spec, steps, dt, "const", mode)
    got = _run_product_trace(spec, steps, dt, "const", mode, path=path)
    if path in ("turnstile", "turnstile_reread") and 512 < spec[1] <= 8192:
        # learn_iter AND learn_vec: one launch per vector step, rows handed over inside it (qe_step_turn.h)
        from dist_classicrl_amd import _lib
        assert _lib.decode_variant(got["stats"]["kernel_variant"])["path"] == "turnstile"
    assert np.array_equal(got["actions"], want["actions"])
    assert np.array_equal(got["q"], want["q"])
    assert np.array_equal(got["history"], want["history"])
    assert np.array_equal(got["final_obs"], want["final_obs"])
    assert np.array_equal(got["agent_rewards"], want["agent_rewards"])


@pytest.mark.parametrize("path", ["turnstile", "turnstile_reread"])
@pytest.mark.parametrize("mode", ["iter", "vec"])
@pytest.mark.parametrize("dt", ["f4", "f8"])
@pytest.mark.parametrize(("agents", "rows"), [(18, 2), (20, 2), (22, 2), (27, 3), (33, 3), (40, 4), (64, 5), (200, 7)])
def test_turnstile_row_records_around_their_capacity(agents, rows, dt, mode, path):
    """A row's record holds ten touchers (TURN_ENTRIES, qe_kernels.h); the eleventh onwards goes on the overflow list.
    Tiny tables put 9 .. 60 touchers on every row -- records exactly full, one over, and mostly on the list -- and the
    path must still equal the reference's sequential order."""
    spec = ("hash", agents, rows, 4, False)
    want = run_oracle_trace(spec, 40, dt, "bench", mode)
    got = _run_product_trace(spec, 40, dt, "bench", mode, path=path)
    for k in ("actions", "q", "history", "final_obs", "agent_rewards", "final_sched"):
        assert np.array_equal(got[k], want[k], equal_nan=(k == "q")), k
    from dist_classicrl_amd import _lib

    assert _lib.decode_variant(got["stats"]["kernel_variant"])["path"] == "turnstile"


@pytest.mark.parametrize("bits", [4, 9, 14])
@pytest.mark.parametrize("path", ["stepwise", "wide", "wide_listed"])
@pytest.mark.parametrize(("spec", "steps", "dt", "mode"), [
    (("hash", 1024, 5000, 16, False), 30, "f4", "iter"),
    (("hash", 4096, 300000, 8, False), 12, "f4", "iter"),
    (("hash", 3000, 40000, 16, False), 10, "f8", "iter"),
    (("hash", 2500, 40000, 16, True), 12, "f4", "vec"),
    (("ttt", 700), 30, "f8", "iter"),
])
def test_hashed_touch_counters_change_nothing(spec, steps, dt, mode, path, bits):
    """QE_OPT_STAMP_HASH_BITS: the touch counters of the step-wise / wide kernels in 2^bits hashed slots (automatic for
    tables of more than 2^22 rows).  Rows that collide in the hash count as shared and take the ordered path, which keys on
    the rows themselves -- with 16 slots nearly every agent does: results must not move."""
    from dist_classicrl_amd import _lib

    want = run_oracle_trace(spec, steps, dt, "const", mode)
    Algo, Runtime, _, _ = _product()
    env = make_device_env(spec)
    algo = Algo(env.state_size, env.action_size, 0.99, seed=0, dtype=np.dtype(dt))
    algo.set_engine_option(_lib.OPT_STAMP_HASH_BITS, bits)
    if path == "wide_listed":
        algo.set_rollout_path("wide")
        algo.set_engine_option(_lib.OPT_LISTED_MIN_AGENTS, 1)
        algo.set_engine_option(_lib.OPT_TOKEN_ROUNDS, 7)
    else:
        algo.set_rollout_path(path)
    lr_p, eps_p = schedule_params("const")
    rt = Runtime(algo, make_schedule(lr_p), make_schedule(eps_p), learn_mode=mode)
    rt.trace_actions = True
    _avg, history, _env, sd = rt.run_steps(steps, env, None)
    assert np.array_equal(rt.last_trace, want["actions"])
    assert np.array_equal(np.asarray(algo.q_table), want["q"])
    assert np.array_equal(np.array(history, dtype=np.float32), want["history"])
    # ... and the batch API, whose kernels count touches in the same slots
    from oracle.qlearn_oracle import OracleQLearning

    rng = np.random.default_rng(bits)
    S, A, n = env.state_size, env.action_size, 900
    batch = (rng.integers(0, min(S, 50), n), rng.integers(0, A, n), rng.random(n).astype(np.float32),
             rng.integers(0, min(S, 50), n), rng.random(n) < 0.1)
    ref = OracleQLearning(S, A, 0.99, seed=0, dtype=np.dtype(dt))
    ref.q_table = np.array(want["q"])
    ref.learn(*[np.array(b) for b in batch], 0.1)
    algo.learn(*batch, 0.1)
    assert np.array_equal(np.asarray(algo.q_table), ref.q_table)


@pytest.mark.parametrize("path", ["stepwise", "wide"])
def test_large_table_takes_hashed_counters_automatically(path):
    """More than 2^22 rows: 2^21 hashed counter slots without being asked (the 16-B-per-row counter array of a 1e7-row
    table would not stay in the Infinity Cache); bit-exact against the C oracle."""
    from oracle import c_oracle

    n, S, A, steps = 3000, 5_000_000, 16, 40
    Algo, Runtime, envs, sch = _product()
    algo = Algo(S, A, 0.99, seed=0)
    algo.set_rollout_path(path)
    rt = Runtime(algo, sch.ExponentialSchedule(0.1, 1e-5, 0.995), sch.ExponentialSchedule(1.0, 0.01, 0.995))
    rt.trace_actions = True
    _avg, history, _, sd = rt.run_steps(steps, envs.HashTabularEnv(n, S, A, seed=1), None)
    ref = c_oracle.CHashRollout(n, S, A, dtype=np.float32)
    eps, _ = c_oracle.exp_schedule(1.0, 0.01, 0.995, n, steps)
    lr, _ = c_oracle.exp_schedule(0.1, 1e-5, 0.995, n, steps)
    want = ref.run(eps, lr, trace=True)
    assert np.array_equal(rt.last_trace, want["actions"])
    assert np.array_equal(np.asarray(algo.q_table), ref.q)
    assert np.array_equal(np.array(history, dtype=np.float32), want["history"])
    assert np.array_equal(sd["states"], ref.obs)


@pytest.mark.parametrize("path", ["stepwise", "persistent", "wide", "turnstile"])
def test_rollout_without_selectable_action_raises_like_the_reference(path):
    """A NaN row maximum leaves `np.where(row == max)` empty and the reference's `random.choice` raises
    IndexError (q_learning_optimal.py:563); the fused rollout reports the same instead of writing outside
    the table."""
    Algo, Runtime, envs, sch = _product()
    n, S, A = (300, 50, 8) if path == "persistent" else (2100, 50, 8)
    algo = Algo(S, A, 0.99, seed=0)
    algo.q_table = np.full((S, A), np.nan, dtype=np.float32)
    algo.set_rollout_path(path)
    rt = Runtime(algo, sch.ConstantSchedule(0.1), sch.ConstantSchedule(0.1))
    with pytest.raises(IndexError, match="empty sequence"):
        rt.run_steps(5, envs.HashTabularEnv(n, S, A, seed=1), None)
    # the reference's mechanism on such a row (choose_actions_vec, q_learning_optimal.py:548-563)
    import random

    row = np.full(A, np.nan, dtype=np.float32)
    candidates = np.where(row == np.max(row))[0]
    assert candidates.size == 0
    with pytest.raises(IndexError):
        random.Random(0).choice(candidates)


def test_rollout_resume_equals_one_shot():
    """run_steps(a) then run_steps(b, curr_state_dict) == run_steps(a+b) (single_thread_runtime.py:58-61)."""
    spec = ("hash", 256, 3000, 16, False)
    one = _run_product_trace(spec, 40, "f4", "bench", "iter", path="stepwise")
    Algo, Runtime, _, _ = _product()
    env = make_device_env(spec)
    algo = Algo(env.state_size, env.action_size, 0.99, seed=0)  # auto -> persistent kernel
    lr_p, eps_p = schedule_params("bench")
    rt = Runtime(algo, make_schedule(lr_p), make_schedule(eps_p))
    _, h1, env, sd = rt.run_steps(15, env, None)
    _, h2, env, sd = rt.run_steps(25, env, sd)
    assert np.array_equal(np.asarray(algo.q_table), one["q"])
    assert np.array_equal(np.array(h1 + h2, dtype=np.float32), one["history"])
    assert np.array_equal(sd["states"], one["final_obs"])


def test_switching_rollout_paths_on_one_engine_equals_one_shot():
    """The turnstile path keeps its row lists in the array the other paths use as touch counters: a run that
    switches turnstile -> step-wise -> wide -> turnstile -> `learn()` batches on ONE engine equals the one-shot run
    (the array is handed over zeroed, list tags of earlier calls never match)."""
    spec = ("hash", 2100, 900, 16, False)
    one = _run_product_trace(spec, 40, "f4", "bench", "iter", path="stepwise")
    Algo, Runtime, _, _ = _product()
    env = make_device_env(spec)
    algo = Algo(env.state_size, env.action_size, 0.99, seed=0)
    lr_p, eps_p = schedule_params("bench")
    rt = Runtime(algo, make_schedule(lr_p), make_schedule(eps_p))
    sd, history = None, []
    for path, k in (("turnstile", 9), ("stepwise", 11), ("wide", 7), ("turnstile", 13)):
        algo.set_rollout_path(path)
        _, h, env, sd = rt.run_steps(k, env, sd)
        history += h
    assert np.array_equal(np.asarray(algo.q_table), one["q"])
    assert np.array_equal(np.array(history, dtype=np.float32), one["history"])
    assert np.array_equal(sd["states"], one["final_obs"])
    # ... and the batch API (its kernels count touches in the same array) still equals the oracle afterwards
    from oracle.qlearn_oracle import OracleQLearning

    rng = np.random.default_rng(5)
    n = 600
    batch = (rng.integers(0, 900, n), rng.integers(0, 16, n), rng.random(n).astype(np.float32),
             rng.integers(0, 900, n), rng.random(n) < 0.1)
    ref = OracleQLearning(900, 16, 0.99, seed=0, dtype=np.dtype("f4"))
    ref.q_table = np.array(one["q"], dtype=np.float32)
    ref.learn(*[np.array(b) for b in batch], 0.1)
    algo.learn(*batch, 0.1)
    assert np.array_equal(np.asarray(algo.q_table), ref.q_table)


# ------------------------------------------------------------------------------- full sizes
@pytest.mark.parametrize(
    ("n", "S", "A", "masked"),
    [(128, 10_000, 8, False), (4096, 1_000_000, 16, False), (1024, 1_000_000, 64, True),
     (128, 1_000_000, 16, False)],
)
def test_full_size_properties(n, S, A, masked):
    """BASELINE configs 2, 3, 5 and the headline shape: properties that do not need the oracle."""
    Algo, Runtime, envs, sch = _product()
    algo = Algo(S, A, 0.99, seed=0)
    env = envs.HashTabularEnv(n, S, A, seed=1, masked=masked)
    rt = Runtime(algo, sch.ConstantSchedule(0.1), sch.ConstantSchedule(0.1))
    steps = 200
    _avg, history, _, sd = rt.run_steps(steps, env, None)
    q = np.asarray(algo.q_table)
    assert np.isfinite(q).all()
    # rewards are in [0, 1): Q is bounded by the geometric series, and never negative
    assert q.min() >= 0.0 and q.max() <= 1.0 / (1.0 - 0.99)
    # at most one cell is written per agent-step
    assert np.count_nonzero(q) <= n * steps
    # episode accounting: every env-step's reward is either in a finished episode or still pending
    obs = sd["states"]["observation"] if masked else sd["states"]
    assert obs.min() >= 0 and obs.max() < S
    assert len(history) == rt.last_stats["episodes"] and rt.last_stats["episodes_dropped"] == 0
    # determinism: the exact-sequential mode is bit-reproducible run to run
    algo2 = Algo(S, A, 0.99, seed=0)
    env2 = envs.HashTabularEnv(n, S, A, seed=1, masked=masked)
    rt2 = Runtime(algo2, sch.ConstantSchedule(0.1), sch.ConstantSchedule(0.1))
    _, history2, _, _ = rt2.run_steps(steps, env2, None)
    assert np.array_equal(np.asarray(algo2.q_table), q)
    assert np.array_equal(np.array(history2), np.array(history))
    # greedy evaluation leaves the table untouched
    val = envs.HashTabularEnv(n, S, A, seed=1, masked=masked)
    rt.evaluate_steps(val, 20 * n)
    assert np.array_equal(np.asarray(algo.q_table), q)


# ------------------------------------------------------------------------------- full sizes, bit-exact
@pytest.mark.parametrize(
    ("name", "n", "S", "A", "masked", "steps"),
    [
        ("headline", 128, 1_000_000, 16, False, 3000),
        ("c2", 128, 10_000, 8, False, 1500),
        ("c3", 4096, 1_000_000, 16, False, 150),
        ("c5", 1024, 1_000_000, 64, True, 100),
        ("c4-one-shard", 8192, 10_000_000, 32, False, 40),
        ("just-resident", 30_000, 200_000, 16, False, 25),  # 469 workgroups: the largest grids the turnstile path takes
        ("not-resident", 40_000, 200_000, 16, False, 20),   # 625 workgroups: back to the wide path
    ],
)
def test_full_size_bit_exact_against_c_oracle(name, n, S, A, masked, steps):
    """BASELINE.json's configurations at full size, benchmark-default schedules: every action of every
    step, the whole Q-table, the episode returns and the final observations must equal the C oracle
    (oracle/qlearn_oracle.c, itself pinned to the NumPy oracle and through it to the reference)."""
    from oracle import c_oracle

    Algo, Runtime, envs, sch = _product()
    algo = Algo(S, A, 0.99, seed=0)
    env = envs.HashTabularEnv(n, S, A, seed=1, masked=masked)
    rt = Runtime(algo, sch.ExponentialSchedule(0.1, 1e-5, 0.995), sch.ExponentialSchedule(1.0, 0.01, 0.995))
    rt.trace_actions = True
    _avg, history, _, sd = rt.run_steps(steps, env, None)
    ref = c_oracle.CHashRollout(n, S, A, masked=masked, dtype=np.float32)
    eps, _ = c_oracle.exp_schedule(1.0, 0.01, 0.995, n, steps)
    lr, _ = c_oracle.exp_schedule(0.1, 1e-5, 0.995, n, steps)
    want = ref.run(eps, lr, trace=True)
    assert np.array_equal(rt.trace_actions, want["actions"])
    got_q = np.asarray(algo.q_table)
    assert np.array_equal(got_q, ref.q)
    assert np.array_equal(np.array(history, dtype=np.float32), want["history"])
    obs = sd["states"]["observation"] if masked else sd["states"]
    assert np.array_equal(obs, ref.obs)
    assert np.array_equal(sd["rewards"], ref.acc)
