"""CPU: the oracle restatement must reproduce every golden vector produced by the real reference
(``tests/golden/make_golden.py``) bit for bit -- this is what pins the oracle."""

import numpy as np
import pytest

from oracle.draws import InjectedDraws, philox4x32
from oracle.qlearn_oracle import OracleQLearning

from helpers import GOLDEN, TRACE_CASES, dense_from_sparse, run_oracle_trace
from golden.make_golden_cases import LEARN_CASES, SELECT_CASES


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [
        ((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         "d16cfe09 94fdcceb 5001e420 24126ea1"),
    ]
    for ctr, key, want in kat:
        got = " ".join(f"{int(x):08x}" for x in philox4x32(*ctr, *key))
        assert got == want


@pytest.mark.parametrize("k", range(len(SELECT_CASES)))
def test_select_matches_reference(k):
    g = np.load(GOLDEN / "select.npz")
    method = SELECT_CASES[k][0]
    S, A, n, masked, det, seed = (int(v) for v in g[f"c{k}_meta"])
    eps, step = float(g[f"c{k}_eps"][0]), int(g[f"c{k}_step"][0])
    algo = OracleQLearning(S, A, 0.9, dtype=g[f"c{k}_q"].dtype)
    algo.q_table = g[f"c{k}_q"].copy()
    algo._rng = algo._np_rng = shim = InjectedDraws(seed)
    masks = g[f"c{k}_masks"].astype(np.int32) if masked else None
    shim.begin(step, n, eps, deterministic=bool(det))
    fn = getattr(algo, method)
    if method == "choose_actions_vec":
        acts = fn(g[f"c{k}_states"], eps, deterministic=bool(det))
    elif method == "choose_masked_actions_vec":
        acts = fn(g[f"c{k}_states"], masks, eps, deterministic=bool(det))
    else:
        acts = fn(g[f"c{k}_states"], eps, deterministic=bool(det), action_masks=masks)
    assert np.array_equal(acts, g[f"c{k}_actions"])


@pytest.mark.parametrize("k", range(len(LEARN_CASES)))
@pytest.mark.parametrize("fn", ["learn", "learn_vec"])
def test_learn_matches_reference(k, fn):
    g = np.load(GOLDEN / "learn.npz")
    S, A, n, masked = (int(v) for v in g[f"c{k}_meta"])
    lr, gamma = (float(v) for v in g[f"c{k}_hyper"])
    algo = OracleQLearning(S, A, gamma, dtype=g[f"c{k}_q0"].dtype)
    algo.q_table = g[f"c{k}_q0"].copy()
    masks = g[f"c{k}_masks"].astype(np.int32) if masked else None
    getattr(algo, fn)(g[f"c{k}_states"], g[f"c{k}_actions"], g[f"c{k}_rewards"],
                      g[f"c{k}_next_states"], g[f"c{k}_terminated"], lr, masks)
    want = g[f"c{k}_q_{fn}"]
    assert algo.q_table.dtype == want.dtype
    assert np.array_equal(algo.q_table, want)  # bit-exact


@pytest.mark.parametrize("name", list(TRACE_CASES))
def test_closed_loop_trace_matches_reference(name):
    g = np.load(GOLDEN / "traces.npz")
    spec, steps, dt, sched, mode = TRACE_CASES[name]
    got = run_oracle_trace(spec, steps, dt, sched, mode)
    assert np.array_equal(got["actions"], g[f"{name}/actions"])
    assert np.array_equal(got["eps"], g[f"{name}/eps"])
    assert np.array_equal(got["lr"], g[f"{name}/lr"])
    want_q = dense_from_sparse(g[f"{name}/q_idx"], g[f"{name}/q_val"], got["q"].shape, got["q"].dtype)
    assert np.array_equal(got["q"], want_q)
    assert np.array_equal(got["history"], g[f"{name}/history"])
    assert np.array_equal(got["final_obs"], g[f"{name}/final_obs"])
    assert np.array_equal(got["agent_rewards"], g[f"{name}/agent_rewards"])
    assert np.array_equal(got["final_sched"], g[f"{name}/final_sched"])


def test_agents_without_valid_action_match_reference():
    """tests/golden/empty_mask.npz (real reference, agent 0 with an all-zero mask): -1 from the list
    variants, a uniform pick over ALL actions from the greedy NumPy variants, IndexError when such an
    agent explores."""
    g = np.load(GOLDEN / "empty_mask.npz")
    S, A, seed, step = (int(v) for v in g["meta"])
    for k in range(int(g["count"])):
        n, det, eps100 = (int(v) for v in g[f"c{k}_cfg"])
        method, eps = str(g[f"c{k}_method"]), eps100 / 100.0
        algo = OracleQLearning(S, A, 0.9)
        algo.q_table = g["q"].copy()
        algo._rng = algo._np_rng = shim = InjectedDraws(seed)
        shim.begin(step, n, eps, deterministic=bool(det))
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


# ------------------------------------------------------------------------------- round 3 goldens
from golden.make_golden_cases import (  # noqa: E402
    NAN_LEARN_CASES, NAN_SELECT_CASES, NAN_TRACE_CASES, SCALE_TRACE_CASES)
from helpers import golden_nan_trace, nan_table, run_oracle_chunks  # noqa: E402


@pytest.mark.parametrize("case", SCALE_TRACE_CASES, ids=[c[0] for c in SCALE_TRACE_CASES])
def test_scale_trace_matches_reference(case):
    """SURVEY 8(c)'s "shrunk C3" (4096 agents on the 1e6 x 16 table) and "C5-small" (1024 masked agents, 64
    actions), generated from the real reference (tests/golden/make_golden_r3.py)."""
    name, spec, steps, dt, sched, learn_fn = case
    g = np.load(GOLDEN / "traces_scale.npz")
    got = run_oracle_trace(spec, steps, dt, sched, "iter" if learn_fn == "learn" else "vec")
    assert np.array_equal(got["actions"], g[f"{name}/actions"].astype(np.int32))
    idx = np.cumsum(g[f"{name}/q_idx_delta"].astype(np.int64))
    want_q = dense_from_sparse(idx, g[f"{name}/q_val"], got["q"].shape, got["q"].dtype)
    assert np.array_equal(got["q"], want_q)
    assert np.array_equal(got["history"], g[f"{name}/history"])
    assert np.array_equal(got["final_obs"], g[f"{name}/final_obs"])
    assert np.array_equal(got["agent_rewards"], g[f"{name}/agent_rewards"])
    assert np.array_equal(got["final_sched"], g[f"{name}/final_sched"])


@pytest.mark.parametrize("k", range(len(NAN_SELECT_CASES)))
def test_selection_on_nan_rows_matches_reference(k):
    """List variants step over a NaN column, NumPy variants take np.max (NaN) and raise IndexError."""
    method, S, A, n, masked, eps, det, dt, cells, tseed = NAN_SELECT_CASES[k]
    g = np.load(GOLDEN / "nan_regime.npz")
    algo = OracleQLearning(S, A, 0.9, dtype=np.dtype(dt))
    algo.q_table = nan_table(S, A, dt, cells, tseed)
    algo._rng = algo._np_rng = shim = InjectedDraws(300 + k)
    shim.begin(11 * k, n, eps, deterministic=det)
    states = g[f"s{k}_states"]
    masks = g[f"s{k}_masks"].astype(np.int32) if masked else None
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
def test_learn_on_nan_rows_matches_reference(k, fn):
    S, A, n, masked, dt, lr, gamma, cells, tseed = NAN_LEARN_CASES[k]
    g = np.load(GOLDEN / "nan_regime.npz")
    algo = OracleQLearning(S, A, gamma, dtype=np.dtype(dt))
    algo.q_table = nan_table(S, A, dt, cells, tseed)
    masks = g[f"l{k}_masks"] if masked else None
    with np.errstate(all="ignore"):
        getattr(algo, fn)(g[f"l{k}_states"], g[f"l{k}_actions"], g[f"l{k}_rewards"], g[f"l{k}_next_states"],
                          g[f"l{k}_terminated"], lr, masks)
    want = g[f"l{k}_q_{fn}"]
    assert np.isnan(want).sum() > cells // 2
    assert np.array_equal(algo.q_table, want, equal_nan=True)


@pytest.mark.parametrize("case", NAN_TRACE_CASES, ids=[c[0] for c in NAN_TRACE_CASES])
def test_closed_loop_through_nan_matches_reference(case):
    """Diverging runs and tables seeded with NaN, cut into run_steps-sized chunks: same tables (NaNs included) at
    every chunk end, IndexError in the same chunk as the real reference."""
    name, spec, chunks, dt, sched, learn_fn, cells, tseed = case
    g = np.load(GOLDEN / "nan_regime.npz")
    want, base = golden_nan_trace(g, name, spec, dt, cells, tseed)
    got = run_oracle_chunks(spec, chunks, dt, sched, "iter" if learn_fn == "learn" else "vec", q0=base if cells else None)
    assert [x["raised"] for x in got] == [x["raised"] for x in want]
    for a, b in zip(got, want, strict=True):
        if b["raised"]:
            continue
        assert np.array_equal(a["actions"], b["actions"][: len(a["actions"])])
        assert np.array_equal(a["q"], b["q"], equal_nan=True)
        assert np.array_equal(a["history"], b["history"], equal_nan=True)
        assert np.array_equal(a["final_obs"], b["final_obs"])
        assert np.array_equal(a["agent_rewards"], b["agent_rewards"], equal_nan=True)
