"""GPU: a Q-table that holds NaN (diverged training, or uploaded that way).

The reference takes ``np.max`` of a row for every TD target (``q_learning_optimal.py:757-761``, ``:884-888``) and in
its NumPy selection variants (``:428``, ``:466``, ``:548``, ``:616``): a NaN in a valid column makes the maximum NaN,
the target NaN, and a greedy pick impossible (``random.choice([])`` -> IndexError, ``:430``, ``:470``, ``:563``).
Its list variants (fewer than 100 unmasked agents, or masked with at most 10 actions, ``:644-726``) scan with
``if v > max_val`` and step over NaN columns (``:290-296``, ``:337-344``).  The kernels follow both: tables must be
equal to the oracle's INCLUDING their NaNs (``equal_nan``), and the IndexError must come in the same ``run_steps``
call in which the reference raises it.
"""

import numpy as np
import pytest

from helpers import run_oracle_chunks, schedule_params

pytestmark = pytest.mark.gpu

from test_gpu_parity import _product, make_device_env, make_schedule  # noqa: E402


def _nan_table(S, A, dt, cells, seed=3):
    """Random table with `cells` NaN entries."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((S, A)).astype(dt)
    q.ravel()[rng.choice(S * A, size=cells, replace=False)] = np.nan
    return q


def _run_product_chunks(spec, chunks, dt, sched, mode, path, q0=None, trace=True, gamma=0.99, seed=0):
    Algo, Runtime, _, _ = _product()
    env = make_device_env(spec)
    algo = Algo(env.state_size, env.action_size, gamma, seed=seed, dtype=np.dtype(dt))
    if q0 is not None:
        algo.q_table = np.array(q0, dtype=np.dtype(dt))
    algo.set_rollout_path(path)
    lr_p, eps_p = schedule_params(sched)
    rt = Runtime(algo, make_schedule(lr_p), make_schedule(eps_p), learn_mode=mode)
    out, sd, history, actions = [], None, [], []
    for k in chunks:
        if trace:
            rt.trace_actions = True
        try:
            try:
                _avg, h, env, sd = rt.run_steps(k, env, sd)
            except ZeroDivisionError:
                h, sd = [], env.state_dict()
        except IndexError:
            out.append({"raised": True})
            break
        history += h
        if trace:
            actions.append(rt.last_trace)
        obs = sd["states"]["observation"] if isinstance(sd["states"], dict) else sd["states"]
        out.append({"raised": False, "q": np.asarray(algo.q_table).copy(), "history": np.array(history, dtype=np.float32),
                    "final_obs": np.array(obs), "agent_rewards": np.array(sd["rewards"]),
                    "actions": np.concatenate(actions) if trace else None})
    return out


def _compare(got, want):
    assert len(got) == len(want), ([g["raised"] for g in got], [w["raised"] for w in want])
    for g, w in zip(got, want, strict=True):
        assert g["raised"] == w["raised"]  # IndexError in the same call as the reference
        if w["raised"]:
            continue
        if g["actions"] is not None:
            assert np.array_equal(g["actions"], w["actions"])
        assert np.array_equal(g["q"], w["q"], equal_nan=True)
        assert np.array_equal(g["history"], w["history"], equal_nan=True)
        assert np.array_equal(g["final_obs"], w["final_obs"])
        assert np.array_equal(g["agent_rewards"], w["agent_rewards"], equal_nan=True)


DIVERGING = [
    # spec, chunks, dtype, mode                                    (lr = 1, colliding increments: overflow -> inf - inf)
    (("bandit", 600, 5), [10] * 6, "f4", "vec"),    # the configuration of the round-2 sweep that finished without an error
    (("hash", 600, 4, 4, False), [8] * 6, "f4", "vec"),
    (("hash", 300, 6, 8, False), [8] * 6, "f4", "vec"),
    (("hash", 2100, 3, 8, False), [6] * 6, "f4", "vec"),
    (("hash", 256, 5, 16, True), [8] * 6, "f4", "vec"),   # masked, A > 10: NumPy variants
    (("hash", 200, 2, 4, False), [80] * 4, "f8", "vec"),
]


@pytest.mark.parametrize("path", ["stepwise", "persistent", "wide", "auto"])
@pytest.mark.parametrize(("spec", "chunks", "dt", "mode"), DIVERGING)
def test_diverging_training_follows_the_reference_into_nan(spec, chunks, dt, mode, path):
    if path == "persistent" and spec[1] > 512:
        pytest.skip("more than 512 agents: the persistent kernel does not apply")
    want = run_oracle_chunks(spec, chunks, dt, "nan", mode)
    assert want[-1]["raised"] or np.isnan(want[-1]["q"]).any(), "the case is meant to reach the NaN regime"
    got = _run_product_chunks(spec, chunks, dt, "nan", mode, path)
    _compare(got, want)


SEEDED = [
    # spec, chunks, dtype, mode, schedule, NaN cells, table seed        (what the oracle does is noted; the test asks it)
    (("hash", 200, 50, 8, False), [10, 10, 10], "f4", "iter", "explore", 40, 3),  # NumPy shape, never greedy: NaN spreads, no error
    (("hash", 200, 2000, 8, False), [3, 3, 3, 3], "f4", "iter", "const", 4, 5),   # ... greedy picks: IndexError in the 2nd call
    (("hash", 1024, 30000, 16, False), [3, 3, 3, 3], "f4", "iter", "const", 6, 5),  # NaN spreads, IndexError in the 3rd call
    (("hash", 1024, 30000, 16, False), [3, 3, 3, 3], "f4", "iter", "const", 6, 3),  # ... in the 4th
    (("hash", 2500, 90000, 16, False), [2, 2, 2, 2], "f4", "iter", "const", 2, 3),
    (("hash", 64, 50, 8, False), [4, 4, 4], "f4", "iter", "const", 6, 4),       # fewer than 100 agents: list variants step over NaN
    (("hash", 90, 400, 16, False), [6, 6], "f8", "iter", "const", 60, 3),
    (("hash", 128, 300, 8, True), [5, 5], "f4", "iter", "const", 20, 4),        # masked, A <= 10: list variants
    (("hash", 128, 3000, 16, True), [3, 3, 3, 3], "f4", "iter", "const", 6, 3),  # masked, A > 10: NumPy variants, 3rd call raises
    (("hash", 300, 80, 8, False), [5, 5], "f8", "vec", "explore", 60, 3),
    (("ttt", 128), [10, 10], "f4", "iter", "const", 300, 3),                    # A = 9 masked: list variants
]


@pytest.mark.parametrize("path", ["stepwise", "persistent", "wide", "turnstile", "auto"])
@pytest.mark.parametrize(("spec", "chunks", "dt", "mode", "sched", "cells", "tseed"), SEEDED)
def test_tables_that_hold_nan_follow_the_reference(spec, chunks, dt, mode, sched, cells, tseed, path):
    if path == "persistent" and spec[1] > 512:
        pytest.skip("more than 512 agents: the persistent kernel does not apply")
    S, A = (19683, 9) if spec[0] == "ttt" else (spec[2], spec[3])
    q0 = _nan_table(S, A, dt, cells, tseed)
    want = run_oracle_chunks(spec, chunks, dt, sched, mode, q0=q0)
    ok = [w for w in want if not w["raised"]]
    if ok:  # (the list variants return -1 for a row without a usable column; the cases stay clear of that)
        assert (ok[-1]["actions"] >= 0).all()
    got = _run_product_chunks(spec, chunks, dt, sched, mode, path, q0=q0)
    _compare(got, want)


@pytest.mark.parametrize("ordered_path", [1, 2, 3])
@pytest.mark.parametrize(("spec", "chunks", "sched", "cells", "tseed"), [
    (("hash", 128, 2000, 16, False), [20, 20, 20, 20], "const", 1, 4),   # IndexError in the 3rd call
    (("hash", 128, 2000, 16, False), [64, 64], "explore", 1600, 3),
    (("hash", 64, 500, 8, False), [30, 30], "const", 10, 4),             # 64 agents: list variants
    (("hash", 128, 60, 16, False), [10, 10, 10], "explore", 50, 3),      # contested steps of the light / full builds
])
def test_nan_tables_through_the_untraced_lean_builds(spec, chunks, sched, cells, tseed, ordered_path):
    from dist_classicrl_amd import _lib

    q0 = _nan_table(spec[2], spec[3], "f4", cells, tseed)
    want = run_oracle_chunks(spec, chunks, "f4", sched, "iter", q0=q0)
    Algo, Runtime, _, _ = _product()
    env = make_device_env(spec)
    algo = Algo(env.state_size, env.action_size, 0.99, seed=0)
    algo.q_table = q0
    algo.set_engine_option(_lib.OPT_LANE_ORDERED_PATH, ordered_path)
    lr_p, eps_p = schedule_params(sched)
    rt = Runtime(algo, make_schedule(lr_p), make_schedule(eps_p))
    sd, history = None, []
    for k, w in zip(chunks, want):
        if w["raised"]:
            with pytest.raises(IndexError):
                rt.run_steps(k, env, sd)
            break
        try:
            _avg, h, env, sd = rt.run_steps(k, env, sd)
        except ZeroDivisionError:
            h, sd = [], env.state_dict()
        d = _lib.decode_variant(rt.last_stats["kernel_variant"])
        assert d["path"] == "persistent" and d["lean"] == 1, d
        history += h
        assert np.array_equal(np.asarray(algo.q_table), w["q"], equal_nan=True)
        assert np.array_equal(np.array(history, dtype=np.float32), w["history"], equal_nan=True)
        assert np.array_equal(sd["states"], w["final_obs"])


# ------------------------------------------------------------------------------- against the real reference's vectors
from golden.make_golden_cases import NAN_LEARN_CASES, NAN_SELECT_CASES, NAN_TRACE_CASES  # noqa: E402
from helpers import GOLDEN, golden_nan_trace, nan_table  # noqa: E402


@pytest.mark.parametrize("k", range(len(NAN_SELECT_CASES)))
def test_selection_on_nan_rows_matches_reference_golden(k):
    method, S, A, n, masked, eps, det, dt, cells, tseed = NAN_SELECT_CASES[k]
    Algo = _product()[0]
    g = np.load(GOLDEN / "nan_regime.npz")
    algo = Algo(S, A, 0.9, seed=300 + k, dtype=np.dtype(dt))
    algo.q_table = nan_table(S, A, dt, cells, tseed)
    algo.step_counter = 11 * k
    states = g[f"s{k}_states"]
    masks = g[f"s{k}_masks"] if masked else None
    try:
        if method == "choose_actions_vec":
            got = algo.choose_actions_vec(states, eps, deterministic=det)
        elif method == "choose_masked_actions_vec":
            got = algo.choose_masked_actions_vec(states, masks, eps, deterministic=det)
        else:
            got = getattr(algo, method)(states, eps, deterministic=det, action_masks=masks)
        raised = 0
    except IndexError:
        got, raised = np.zeros(0, dtype=np.int32), 1
    assert raised == int(g[f"s{k}_raised"][0])
    assert np.array_equal(np.asarray(got, dtype=np.int32), g[f"s{k}_actions"])


@pytest.mark.parametrize("k", range(len(NAN_LEARN_CASES)))
@pytest.mark.parametrize("fn", ["learn", "learn_vec"])
def test_learn_on_nan_rows_matches_reference_golden(k, fn):
    S, A, n, masked, dt, lr, gamma, cells, tseed = NAN_LEARN_CASES[k]
    Algo = _product()[0]
    g = np.load(GOLDEN / "nan_regime.npz")
    algo = Algo(S, A, gamma, seed=0, dtype=np.dtype(dt))
    algo.q_table = nan_table(S, A, dt, cells, tseed)
    masks = g[f"l{k}_masks"] if masked else None
    getattr(algo, fn)(g[f"l{k}_states"], g[f"l{k}_actions"], g[f"l{k}_rewards"], g[f"l{k}_next_states"],
                      g[f"l{k}_terminated"], lr, masks)
    assert np.array_equal(np.asarray(algo.q_table), g[f"l{k}_q_{fn}"], equal_nan=True)


@pytest.mark.parametrize("path", ["stepwise", "persistent", "wide", "turnstile", "auto"])
@pytest.mark.parametrize("case", NAN_TRACE_CASES, ids=[c[0] for c in NAN_TRACE_CASES])
def test_closed_loop_through_nan_matches_reference_golden(case, path):
    name, spec, chunks, dt, sched, learn_fn, cells, tseed = case
    mode = "iter" if learn_fn == "learn" else "vec"
    if path == "persistent" and spec[1] > 512:
        pytest.skip("more than 512 agents: the persistent kernel does not apply")
    g = np.load(GOLDEN / "nan_regime.npz")
    want, base = golden_nan_trace(g, name, spec, dt, cells, tseed)
    got = _run_product_chunks(spec, chunks, dt, sched, mode, path, q0=base if cells else None)
    assert [x["raised"] for x in got] == [x["raised"] for x in want]
    for a, b in zip(got, want, strict=True):
        if b["raised"]:
            continue
        assert np.array_equal(a["actions"], b["actions"][: len(a["actions"])])
        assert np.array_equal(a["q"], b["q"], equal_nan=True)
        assert np.array_equal(a["history"], b["history"], equal_nan=True)
        assert np.array_equal(a["final_obs"], b["final_obs"])
        assert np.array_equal(a["agent_rewards"], b["agent_rewards"], equal_nan=True)
