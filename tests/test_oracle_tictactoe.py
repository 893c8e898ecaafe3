"""CPU: the oracle's TicTacToe rules against the known answers the reference's own environment tests
hold (/root/reference/tests/dist_classicrl/environment/test_tiktaktoe_mod.py: winner detection
:11-39, valid moves :42-69, observation / mask :72-118, apply-move outcomes :150-264), restated on
occupancy masks, plus the mixed-radix observation id (utils.py:12-48)."""

import numpy as np

from oracle.envs import TicTacToeVecEnv, ttt_encode, ttt_winner


def masks(board):
    flat = np.asarray(board).ravel()
    return (int(sum(1 << i for i in range(9) if flat[i] == 1)), int(sum(1 << i for i in range(9) if flat[i] == 2)))


def test_check_winner_known_answers():
    assert ttt_winner(*masks(np.zeros((3, 3)))) is None
    assert ttt_winner(*masks([[1, 1, 1], [0, 2, 0], [0, 0, 2]])) == 1
    assert ttt_winner(*masks([[2, 2, 0], [0, 2, 0], [0, 2, 1]])) == 2
    assert ttt_winner(*masks([[1, 0, 2], [1, 1, 2], [0, 0, 1]])) == 1
    assert ttt_winner(*masks([[1, 2, 1], [1, 2, 2], [2, 1, 1]])) is None  # full board, draw


def test_observation_id_and_action_mask():
    env = TicTacToeVecEnv(1)
    radix = 3 ** np.arange(8, -1, -1)  # compute_radix([3] * 9)
    for board in ([[0] * 3] * 3, [[1, 0, 0], [0, 0, 0], [0, 0, 0]], [[1, 2, 2], [1, 1, 2], [2, 2, 1]],
                  [[1, 0, 0], [2, 0, 0], [0, 0, 0]]):
        flat = np.asarray(board).ravel()
        sid = ttt_encode(*masks(board))
        assert sid == int(np.dot(flat, radix))  # encode_multi_discrete
        assert np.array_equal(env.action_masks(np.array([sid]))[0], (flat == 0).astype(np.int8))


def _set(env, board, agent_mark=1):
    env.m1[0], env.m2[0] = masks(board)
    env.agent_mark[0] = agent_mark


def test_apply_move_outcomes():
    env = TicTacToeVecEnv(1)
    env.reset()
    # agent completes a row -> +1, terminated, next observation is a fresh episode
    _set(env, [[1, 1, 0], [2, 2, 0], [0, 0, 0]])
    obs, r, te, tr, _ = env.step(np.array([2]))
    assert r[0] == 1.0 and te[0] and not tr[0]
    assert bin(int(env.m1[0] | env.m2[0])).count("1") <= 1  # empty board, or the machine's opening move
    # agent fills the last cell without a line -> draw, reward 0
    _set(env, [[1, 2, 1], [1, 2, 2], [2, 1, 0]])
    obs, r, te, _, _ = env.step(np.array([8]))
    assert r[0] == 0.0 and te[0]
    # one empty cell left after the agent's move and the machine wins there -> -1
    _set(env, [[2, 2, 0], [1, 1, 0], [1, 0, 2]])
    obs, r, te, _, _ = env.step(np.array([7]))  # agent does not block; machine must play 2 or 5
    assert te[0] == (r[0] == -1.0)
    # the machine's reply is always an empty cell and the game continues otherwise
    _set(env, np.zeros((3, 3)))
    obs, r, te, _, _ = env.step(np.array([4]))
    assert not te[0] and r[0] == 0.0
    assert bin(int(env.m1[0])).count("1") == 1 and bin(int(env.m2[0])).count("1") == 1 and not (env.m1[0] & env.m2[0])


def test_random_games_are_legal_and_balanced():
    env = TicTacToeVecEnv(64, seed=3)
    obs, _ = env.reset()
    rng = np.random.default_rng(0)
    starts = []
    outcomes = {1.0: 0, -1.0: 0, 0.0: 0}
    for _ in range(200):
        mask = obs["action_mask"]
        assert mask.sum(axis=1).min() >= 1
        a = np.array([rng.choice(np.flatnonzero(m)) for m in mask])
        obs, r, te, _, _ = env.step(a)
        for x in r[te]:
            outcomes[float(x)] += 1
        starts.extend((env.agent_mark[te] == 1).tolist())
    assert min(outcomes.values()) > 0  # wins, losses and draws all occur
    assert 0.35 < np.mean(starts) < 0.65  # who starts is a fair coin
