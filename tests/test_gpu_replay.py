"""GPU: the device-resident ExperienceReplay against the real reference's fixture and, for the device-side
`learn_from`, against sampling + `learn` done by the oracle."""

import numpy as np
import pytest

from test_oracle_replay import replay_script

pytestmark = pytest.mark.gpu


def _classes():
    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.buffers import ExperienceReplay

    return ExperienceReplay, OptimalQLearningBase


def test_replay_matches_reference_fixture():
    replay_script(_classes()[0])


def test_push_batch_wraps_like_repeated_push():
    from oracle.replay_oracle import OracleReplay

    Replay = _classes()[0]
    rng = np.random.default_rng(0)
    for capacity, n in [(10, 3), (10, 10), (10, 27), (64, 500), (7, 7)]:
        rb, ref = Replay(capacity, 1), OracleReplay(capacity, 1)
        for _ in range(3):
            s, a = rng.integers(1 << 40, size=n), rng.integers(100, size=n)
            r, nx, d = rng.standard_normal(n), rng.integers(1 << 40, size=n), rng.random(n) < 0.3
            rb.push_batch(s, a, r, nx, d)
            for e in zip(s, a, r, nx, d):
                ref.push(e)
            assert (rb.position, rb.full, len(rb)) == (ref.position, ref.full, len(ref))
            k = len(ref)  # slots 0..k-1 hold data (the rest of the ring was never written)
            assert np.array_equal(rb.state_buffer[:k], ref.state_buffer[:k])
            assert np.array_equal(rb.action_buffer[:k], ref.action_buffer[:k])
            assert np.array_equal(rb.reward_buffer[:k], ref.reward_buffer[:k])
            assert np.array_equal(rb.next_state_buffer[:k], ref.next_state_buffer[:k])
            assert np.array_equal(rb.done_buffer[:k], ref.done_buffer[:k])
        got, want = rb.sample_arrays(min(5, len(ref))), ref.sample_arrays(min(5, len(ref)))
        assert all(np.array_equal(x, y) for x, y in zip(got, want))


@pytest.mark.parametrize("mode", ["iter", "vec"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_learn_from_replay_on_device_equals_sample_then_learn(mode, dt):
    from oracle.qlearn_oracle import OracleQLearning
    from oracle.replay_oracle import OracleReplay

    Replay, Algo = _classes()
    S, A, capacity, batch = 40, 6, 300, 128
    rng = np.random.default_rng(3)
    rb, ref_rb = Replay(capacity, 9), OracleReplay(capacity, 9)
    n = 450  # wraps
    s, a = rng.integers(S, size=n), rng.integers(A, size=n)
    r, nx, d = rng.random(n).astype(np.float32).astype(np.float64), rng.integers(S, size=n), rng.random(n) < 0.2
    rb.push_batch(s, a, r, nx, d)
    for e in zip(s, a, r, nx, d):
        ref_rb.push(e)
    q0 = rng.standard_normal((S, A)).astype(dt)
    algo, ref = Algo(S, A, 0.9, seed=0, dtype=np.dtype(dt)), OracleQLearning(S, A, 0.9, dtype=np.dtype(dt))
    algo.q_table = q0
    ref.q_table = q0.copy()
    for _ in range(4):
        idx = rb.learn_from(algo, batch, 0.1, mode=mode)
        bs, ba, br, bn, bd = ref_rb.sample_arrays(batch)
        fn = ref.learn if mode == "iter" else ref.learn_vec
        fn(bs.astype(np.int32), ba.astype(np.int32), br.astype(np.float32), bn.astype(np.int32), bd, 0.1)
        assert np.array_equal(np.asarray(algo.q_table), ref.q_table)


def test_replay_errors():
    Replay, Algo = _classes()
    rb = Replay(8, 0)
    rb.push((1, 2, 0.5, 3, False))
    with pytest.raises(ValueError):
        rb.sample(2)  # numpy: cannot take a larger sample than population when replace=False
    with pytest.raises(ValueError):
        rb.push_batch([1, 2], [0], [0.0], [1], [False])
    rb.push((99, 0, 0.0, 1, False))  # a state the table below does not have
    with pytest.raises(IndexError):
        rb.learn_from(Algo(10, 3, 0.9, seed=0), 2, 0.1)


@pytest.mark.parametrize(("n", "S", "A", "steps", "capacity", "path"), [
    (96, 500, 8, 25, 96 * 25 + 17, "auto"),      # persistent kernel, everything fits the ring
    (96, 500, 8, 40, 1000, "auto"),              # ... and wrapping around it
    (600, 3000, 16, 12, 600 * 12, "stepwise"),   # step-wise kernels
    (600, 3000, 16, 12, 600 * 12, "auto"),       # one launch per step (turnstile path)
    (2100, 3000, 16, 9, 5000, "wide"),           # token rounds
])
def test_fused_rollout_pushes_every_transition_into_the_ring(n, S, A, steps, capacity, path):
    """experience_replay.py:68-86 wired to the fused loop: the ring holds (s, a, r, s', done) of every agent and
    vector step in (step, agent) order -- what a host loop pushing after every env.step would store."""
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd.environments import HashTabularEnv
    from dist_classicrl_amd.schedules import ConstantSchedule
    from oracle.envs import HashTabularEnv as OracleEnv

    Replay, Algo = _classes()
    algo = Algo(S, A, 0.99, seed=0)
    algo.set_rollout_path(path)
    rb = Replay(capacity, 1)
    rb.attach(algo)
    rt = GpuRolloutQLearning(algo, ConstantSchedule(0.1), ConstantSchedule(0.3))
    rt.trace_actions = True
    try:
        rt.run_steps(steps, HashTabularEnv(n, S, A, seed=1), None)
    except ZeroDivisionError:
        pass
    actions = rt.last_trace
    # the same environment driven by those actions on the CPU
    env = OracleEnv(n, S, A, seed=1)
    obs, _ = env.reset()
    want = []
    for t in range(steps):
        nxt, r, term, _, _ = env.step(actions[t])
        want.append((obs.copy(), actions[t].copy(), r.copy(), nxt.copy(), term.copy()))
        obs = nxt
    ws, wa, wr, wn, wd = (np.concatenate([w[k] for w in want]) for k in range(5))
    total = steps * n
    assert (rb.position, rb.full, len(rb)) == (total % capacity, total >= capacity, min(total, capacity))
    slots = np.arange(total) % capacity
    live = np.arange(total) >= total - capacity  # later pushes overwrite earlier ones
    gs, ga, gr, gn, gd = rb.state_buffer, rb.action_buffer, rb.reward_buffer, rb.next_state_buffer, rb.done_buffer
    assert np.array_equal(gs[slots[live]], ws[live]) and np.array_equal(ga[slots[live]], wa[live])
    assert np.array_equal(gr[slots[live]], wr[live].astype(np.float64)) and np.array_equal(gn[slots[live]], wn[live])
    assert np.array_equal(gd[slots[live]], wd[live].astype(bool))
    before = np.asarray(algo.q_table).copy()
    rb.learn_from(algo, 64, 0.1)  # the optional replay phase: sample -> learn without leaving the device
    assert not np.array_equal(before, np.asarray(algo.q_table))
    rb.detach(algo)
    pos = rb.position
    try:
        rt.run_steps(3, HashTabularEnv(n, S, A, seed=1), None)
    except ZeroDivisionError:
        pass
    assert rb.position == pos  # detached: nothing is pushed any more
