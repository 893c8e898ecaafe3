"""CPU: the C restatement of the closed loop must agree bit for bit with the NumPy oracle (which is
pinned against the real reference), for both table dtypes, both update semantics and masks."""

import numpy as np
import pytest

from oracle import c_oracle

from helpers import run_oracle_trace


@pytest.mark.parametrize(
    ("n", "S", "A", "masked", "dt", "mode", "steps"),
    [
        (128, 10000, 8, False, "f4", "iter", 40),
        (256, 64, 16, False, "f4", "iter", 30),
        (256, 64, 16, False, "f8", "iter", 30),
        (256, 64, 16, False, "f8", "vec", 30),
        (200, 90, 16, False, "f4", "vec", 30),
        (128, 500, 64, True, "f4", "iter", 30),
        (64, 300, 9, True, "f8", "iter", 30),
        (1, 100, 4, False, "f8", "iter", 60),
    ],
)
def test_c_oracle_matches_numpy_oracle(n, S, A, masked, dt, mode, steps):
    want = run_oracle_trace(("hash", n, S, A, masked), steps, dt, "bench", mode)
    run = c_oracle.CHashRollout(n, S, A, masked=masked, dtype=np.dtype(dt), mode=mode)
    eps, _ = c_oracle.exp_schedule(1.0, 0.01, 0.995, n, steps)
    lr, _ = c_oracle.exp_schedule(0.1, 1e-5, 0.995, n, steps)
    assert np.array_equal(eps, want["eps"]) and np.array_equal(lr, want["lr"])
    got = run.run(eps, lr, trace=True)
    assert np.array_equal(got["actions"], want["actions"])
    assert np.array_equal(run.q, want["q"])
    assert np.array_equal(got["history"], want["history"])
    assert np.array_equal(run.obs, want["final_obs"])
    assert np.array_equal(run.acc, want["agent_rewards"])


@pytest.mark.parametrize(("spec", "chunks", "dt", "mode", "sched", "cells", "tseed"), [
    (("hash", 200, 50, 8, False), [10, 10, 10], "f4", "iter", "explore", 40, 3),   # NumPy selection shape, never greedy
    (("hash", 200, 2000, 8, False), [3, 3, 3, 3], "f4", "iter", "const", 4, 5),    # IndexError in the 2nd chunk
    (("hash", 1024, 30000, 16, False), [3, 3, 3, 3], "f4", "iter", "const", 6, 5),
    (("hash", 64, 50, 8, False), [4, 4, 4], "f4", "iter", "const", 6, 4),          # list variants: NaN columns stepped over
    (("hash", 128, 300, 8, True), [5, 5], "f4", "iter", "const", 20, 4),
    (("hash", 128, 3000, 16, True), [3, 3, 3, 3], "f4", "iter", "const", 6, 3),
    (("hash", 300, 80, 8, False), [5, 5], "f8", "vec", "explore", 60, 3),
    (("hash", 600, 4, 4, False), [8] * 6, "f4", "vec", "nan", 0, 0),               # diverging: inf - inf appears on the way
])
def test_c_oracle_follows_numpy_oracle_through_nan(spec, chunks, dt, mode, sched, cells, tseed):
    """A table that holds NaN: np.max propagates it (TD targets; the NumPy selection variants, whose greedy pick
    then raises IndexError), the list variants' scan steps over it (q_learning_optimal.py:290-296, :548, :757-761)."""
    from helpers import run_oracle_chunks, schedule_params

    _, n, S, A, masked = spec
    rng = np.random.default_rng(tseed)
    q0 = rng.standard_normal((S, A)).astype(dt) if cells else np.zeros((S, A), dtype=dt)
    if cells:
        q0.ravel()[rng.choice(S * A, size=cells, replace=False)] = np.nan
    want = run_oracle_chunks(spec, chunks, dt, sched, mode, q0=q0)
    run = c_oracle.CHashRollout(n, S, A, masked=masked, dtype=np.dtype(dt), mode=mode)
    run.q[:] = q0
    (_, lr_v, _, _), (_, eps_v, _, _) = schedule_params(sched)
    history = []
    for k, w in zip(chunks, want):
        if w["raised"]:
            with pytest.raises(IndexError):
                run.run(np.full(k, eps_v), np.full(k, lr_v), trace=True)
            break
        with np.errstate(all="ignore"):
            got = run.run(np.full(k, eps_v), np.full(k, lr_v), trace=True)
        history.append(got["history"])
        assert np.array_equal(got["actions"], w["actions"][-k:])
        assert np.array_equal(run.q, w["q"], equal_nan=True)
        assert np.array_equal(np.concatenate(history), w["history"], equal_nan=True)
        assert np.array_equal(run.obs, w["final_obs"])
    assert any(w["raised"] for w in want) or np.isnan(want[-1]["q"]).any()
