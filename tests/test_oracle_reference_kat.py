"""CPU: the known-answer cases the reference's own tests hold for the hot path, restated as literals
and run against the oracle.  Sources, relative to ``/root/reference/tests/``:

* ``dist_classicrl/algorithms/base_algorithms/test_q_learning_optimal.py:148-282`` (TD updates,
  duplicate-index divergence 5.0 vs 3.5 and 4.5 vs 3.25), ``:337-633`` (selection);
* ``dist_classicrl/algorithms/runtime/test_q_learning_runtimes.py:77-98`` (bandit run_steps);
* ``dist_classicrl/algorithms/runtime/test_runtime_evals.py:57-116`` (evaluate_steps / _episodes).
"""

from unittest.mock import patch

import numpy as np
import pytest

from oracle.envs import RiggedBanditVecEnv
from oracle.qlearn_oracle import OracleQLearning, OracleRuntime, OracleSchedule

LEARNERS = ["learn", "learn_iter", "learn_vec"]
i32 = lambda *v: np.array(v, dtype=np.int32)  # noqa: E731


@pytest.mark.parametrize("fn", LEARNERS)
def test_td_single_no_mask(fn):
    ql = OracleQLearning(4, 3, discount_factor=0.5)
    ql.q_table[1] = [1.0, 2.0, 0.5]
    getattr(ql, fn)(i32(0), i32(2), np.array([1.0], np.float32), i32(1), np.array([False]), 1.0)
    assert ql.q_table[0, 2] == 2.0  # 1 + 0.5 * 2.0


@pytest.mark.parametrize("fn", LEARNERS)
def test_td_single_masked_suboptimal(fn):
    ql = OracleQLearning(4, 3, discount_factor=0.5)
    ql.q_table[2] = [1.0, 3.0, 2.5]
    getattr(ql, fn)(i32(0), i32(0), np.array([0.0], np.float32), i32(2), np.array([False]), 1.0,
                    np.array([[1, 0, 1]], dtype=np.int32))
    assert ql.q_table[0, 0] == 1.25  # 0.5 * 2.5


@pytest.mark.parametrize("fn", LEARNERS)
@pytest.mark.parametrize("masked", [False, True])
def test_td_duplicates_vec_vs_iter(fn, masked):
    ql = OracleQLearning(5, 3, discount_factor=0.5)
    ql.q_table[2] = [0.5, 1.5, 1.0]
    ql.q_table[3] = [1.0, 3.0, 2.5]
    masks = np.array([[1, 0, 1]] * 3, dtype=np.int32) if masked else None
    getattr(ql, fn)(i32(0, 1, 1), i32(2, 0, 0), np.array([1.0, 0.0, 2.0], np.float32), i32(2, 3, 3),
                    np.array([False] * 3), 1.0, masks)
    if masked:
        assert ql.q_table[0, 2] == 1.5
        assert ql.q_table[1, 0] == (4.5 if fn == "learn_vec" else 3.25)
    else:
        assert ql.q_table[0, 2] == 1.75
        assert ql.q_table[1, 0] == (5.0 if fn == "learn_vec" else 3.5)


BATCH_UNMASKED = ["choose_actions", "choose_actions_iter", "choose_actions_vec_iter", "choose_actions_vec"]
BATCH_MASKED = ["choose_actions", "choose_actions_iter", "choose_actions_vec_iter", "choose_masked_actions_vec"]


def _call(ql, fn, states, eps, det, masks=None):
    if fn == "choose_actions_vec":
        return ql.choose_actions_vec(states, eps, deterministic=det)
    if fn == "choose_masked_actions_vec":
        return ql.choose_masked_actions_vec(states, masks, eps, deterministic=det)
    return getattr(ql, fn)(states, eps, deterministic=det, action_masks=masks)


@pytest.mark.parametrize("fn", BATCH_UNMASKED)
def test_batched_unique_max(fn):
    ql = OracleQLearning(4, 3, 0.9)
    ql.q_table[0] = [0.2, 0.9, 0.1]
    ql.q_table[2] = [0.5, 0.4, 0.7]
    ql._rng.begin(0, 2, 0.0, deterministic=True)
    assert np.array_equal(_call(ql, fn, i32(0, 2), 0.0, True), [1, 2])


@pytest.mark.parametrize("fn", BATCH_MASKED)
def test_batched_masked_unique_max(fn):
    ql = OracleQLearning(4, 3, 0.9)
    ql.q_table[0] = [0.2, 0.9, 0.1]
    ql.q_table[2] = [0.5, 0.4, 0.7]
    ql._rng.begin(0, 2, 0.0, deterministic=True)
    masks = np.array([[1, 0, 1], [1, 1, 0]], dtype=np.int32)
    assert np.array_equal(_call(ql, fn, i32(0, 2), 0.0, True, masks), [0, 0])


@pytest.mark.parametrize("fn", BATCH_UNMASKED)
def test_batched_tie_follows_injected_choice(fn):
    ql = OracleQLearning(3, 3, 0.9)
    ql.q_table[0] = [0.5, 0.5, 0.1]
    ql.q_table[2] = [1.0, 1.0, 0.0]
    with patch.object(ql._rng, "choice", side_effect=[0, 1]):
        assert np.array_equal(_call(ql, fn, i32(0, 2), 0.0, True), [0, 1])


@pytest.mark.parametrize("fn", BATCH_MASKED)
def test_batched_masked_tie_follows_injected_choice(fn):
    ql = OracleQLearning(3, 3, 0.9)
    ql.q_table[0] = [0.7, 0.7, 0.7]
    ql.q_table[1] = [0.1, 0.9, 0.9]
    masks = np.array([[1, 0, 1], [0, 1, 1]], dtype=np.int32)
    with patch.object(ql._rng, "choice", side_effect=[2, 1]):
        assert np.array_equal(_call(ql, fn, i32(0, 1), 0.0, True, masks), [2, 1])


class ForcedExplore:
    """uniform/random -> 0.0 (always explore), randint -> 1, choice prefers 1: the reference's
    ``DeterministicRNG`` behaviour (test_q_learning_runtimes.py:17-45), restated."""

    def uniform(self, _a=0.0, _b=1.0):
        return 0.0

    def random(self):
        return 0.0

    def randint(self, _a, _b):
        return 1

    def choice(self, seq):
        arr = np.asarray(seq)
        return 1 if (arr == 1).any() else int(arr[0])


def test_bandit_run_steps_known_answer():
    algo = OracleQLearning(1, 2, discount_factor=1.0, seed=0)
    algo._rng = ForcedExplore()
    rt = OracleRuntime(algo, OracleSchedule("constant", 1.0), OracleSchedule("linear", 1.0, decay=1.0))
    avg, history, _env, sd = rt.run_steps(5, RiggedBanditVecEnv(1, episode_len=5))
    assert history == [5.0] and avg == 5.0
    assert algo.q_table[0, 1] == 1.0
    assert rt.lr_schedule.get_value() == 1.0
    assert rt.exploration_rate_schedule.get_value() == 6.0
    assert isinstance(sd["states"], np.ndarray)


@pytest.mark.parametrize(("n_envs", "steps"), [(1, 10), (3, 30)])
def test_evaluate_steps_counts_full_episodes(n_envs, steps):
    algo = OracleQLearning(1, 2, 0.99, seed=0)
    algo.q_table[0] = [0.0, 1.0]
    rt = OracleRuntime(algo, OracleSchedule("constant", 0.0), OracleSchedule("constant", 0.0))
    total, history = rt.evaluate_steps(RiggedBanditVecEnv(n_envs, episode_len=10), steps)
    full = (steps // n_envs) // 10 * n_envs
    assert history == [10.0] * full and total == 10.0 * full


@pytest.mark.parametrize(("n_envs", "episodes"), [(1, 3), (4, 8)])
def test_evaluate_episodes(n_envs, episodes):
    algo = OracleQLearning(1, 2, 0.99, seed=0)
    algo.q_table[0] = [0.0, 1.0]
    rt = OracleRuntime(algo, OracleSchedule("constant", 0.0), OracleSchedule("constant", 0.0))
    total, history = rt.evaluate_episodes(RiggedBanditVecEnv(n_envs, episode_len=10), episodes)
    assert history == [10.0] * episodes and total == 10.0 * episodes
