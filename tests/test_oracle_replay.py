"""CPU: the replay-ring oracle against the fixture generated from the real reference class."""

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.replay_oracle import OracleReplay


def replay_script(make):
    """Replays tests/golden/replay.npz against `make(capacity, seed)`; yields nothing, asserts everything."""
    g = np.load(GOLDEN / "replay.npz")
    capacity, seed = (int(v) for v in g["meta"])
    rb = make(capacity, seed)
    pushed = iter(g["pushed"])
    for op, pos, full, length, s, a, r, n, d in g["log"]:
        if op == 0:
            e = next(pushed)
            rb.push((int(e[0]), int(e[1]), float(e[2]), int(e[3]), bool(e[4])))
        else:
            got = rb.sample(1)
            assert got == (int(s), int(a), float(r), int(n), bool(d))
            assert [type(v) for v in got] == [int, int, float, int, bool]
        assert (rb.position, int(rb.full), len(rb)) == (int(pos), int(full), int(length))
    valid = len(rb)
    assert np.array_equal(rb.state_buffer[:valid], g["final_state"])
    assert np.array_equal(rb.action_buffer[:valid], g["final_action"])
    assert np.array_equal(rb.reward_buffer[:valid], g["final_reward"])
    assert np.array_equal(rb.next_state_buffer[:valid], g["final_next"])
    assert np.array_equal(rb.done_buffer[:valid], g["final_done"])
    with pytest.raises(TypeError):  # int(array of two) -- upstream's sample only works for one experience
        rb.sample(2)
    return rb


def test_oracle_replay_matches_reference_fixture():
    replay_script(OracleReplay)
