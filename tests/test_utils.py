"""Host logic: mixed-radix state ids (reference utils.py:12-118)."""

import numpy as np

from dist_classicrl_amd import utils
from oracle.envs import ttt_encode


def test_radix_of_the_tictactoe_board():
    radix = utils.compute_radix(np.array([3] * 9, dtype=np.int32))
    assert radix.dtype == np.int32 and radix.tolist() == [6561, 2187, 729, 243, 81, 27, 9, 3, 1]
    assert utils.compute_radix(np.array([2, 5, 7], dtype=np.int32)).tolist() == [35, 7, 1]
    assert utils.compute_radix(np.array([4], dtype=np.int32)).tolist() == [1]


def test_encode_decode_round_trip_and_device_encoding():
    rng = np.random.default_rng(0)
    nvec = np.array([3] * 9, dtype=np.int32)
    radix = utils.compute_radix(nvec)
    boards = rng.integers(0, 3, size=(200, 9)).astype(np.int32)
    ids = utils.encode_multi_discretes(boards, radix)
    assert ids.min() >= 0 and ids.max() < 3**9
    assert [utils.encode_multi_discrete(b, radix) for b in boards] == ids.tolist()
    assert np.array_equal(utils.decode_to_multi_discretes(nvec, ids[:, None], radix), boards)
    assert np.array_equal(utils.decode_to_multi_discrete(nvec, int(ids[7]), radix), boards[7])
    # the device / oracle environments keep the board as two 9-bit occupancy masks (cell k = bit k) and
    # must produce the same ids: 1 = agent's mark, 2 = opponent's
    for b, want in zip(boards[:50], ids[:50]):
        m1 = sum(1 << k for k in range(9) if b[k] == 1)
        m2 = sum(1 << k for k in range(9) if b[k] == 2)
        assert ttt_encode(m1, m2) == int(want)
