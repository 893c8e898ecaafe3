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
