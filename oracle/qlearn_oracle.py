"""TEST INFRASTRUCTURE ONLY -- interpreted NumPy/Python restatement of the reference hot path.

Structured like the reference on purpose (per-agent Python generators around small NumPy calls) so
that (a) every reference variant has a line-citable counterpart and (b) its single-core speed is a
defensible stand-in for the reference's ``single_thread`` runtime when it is timed as the
``cpu_baseline`` of ``bench.py`` (the reference's own Python cannot travel to the GPU box).

All citations are relative to ``/root/reference/src/dist_classicrl/``.

Randomness: the two generator attributes ``_rng`` (CPython ``random.Random`` surface) and
``_np_rng`` (NumPy ``Generator`` surface) are, by default, one :class:`oracle.draws.InjectedDraws`
object -- the same object the golden-vector generator plants into the real reference.
"""

from __future__ import annotations

import math

import numpy as np

from .draws import InjectedDraws

# dispatcher thresholds, algorithms/base_algorithms/q_learning_optimal.py:14-20
_DET_MAX_A_ITER = 10
_DET_MIN_A_VEC_ITER = 10000
_DET_MAX_N_VEC_ITER = 3
_NOMASK_MAX_N_ITER = 100
_NOMASK_MIN_A_VEC_ITER = 100
_MASK_MAX_A_ITER = 10


class OracleQLearning:
    """Restates ``OptimalQLearningBase`` (algorithms/base_algorithms/q_learning_optimal.py:23-934)."""

    def __init__(self, state_size, action_size, discount_factor=0.97, seed=None, dtype=np.float64):
        # :84-98 -- zero (S, A) table; the reference table is float64, ``dtype=float32`` reproduces
        # the reference after ``algo.q_table = algo.q_table.astype(np.float32)``.
        self.state_size = int(state_size)
        self.action_size = int(action_size)
        self.discount_factor = discount_factor
        self.q_table = np.zeros((self.state_size, self.action_size), dtype=dtype)
        self._rng = self._np_rng = InjectedDraws(0 if seed is None else seed)

    # ------------------------------------------------------------------ selection, scalar forms
    def choose_action(self, state, exploration_rate, *, deterministic=False):
        # :263-302 -- Python scan for the max and its tie list; -1 when nothing is selectable.
        if not deterministic and self._rng.uniform(0, 1) < exploration_rate:
            return self._rng.randint(0, self.action_size - 1)
        best, ties = -math.inf, []
        for j, v in enumerate(self.q_table[state]):
            if v > best:
                best, ties = v, [j]
            elif v == best:
                ties.append(j)
        return self._rng.choice(ties) if ties else -1

    def choose_masked_action(self, state, action_mask, exploration_rate, *, deterministic=False):
        # :304-348 -- same, restricted to mask-valid actions; the exploratory pick is uniform over
        # the valid actions.
        assert len(action_mask) == self.action_size
        if not deterministic and self._rng.uniform(0, 1) < exploration_rate:
            cand = [j for j in range(self.action_size) if action_mask[j]]
        else:
            best, cand = -math.inf, []
            for j, v in enumerate(self.q_table[state]):
                if not action_mask[j]:
                    continue
                if v > best:
                    best, cand = v, [j]
                elif v == best:
                    cand.append(j)
        return self._rng.choice(cand) if cand else self._no_candidate()

    def choose_action_vec(self, state, exploration_rate, *, deterministic=False):
        # :402-430 -- NumPy row max + where(); uses ``_rng.random()`` instead of ``uniform``.
        if not deterministic and self._rng.random() < exploration_rate:
            return self._rng.randint(0, self.action_size - 1)
        row = self.q_table[state]
        return self._rng.choice(np.where(row == np.max(row))[0])

    def choose_masked_action_vec(self, state, action_mask, exploration_rate, *, deterministic=False):
        # :432-470 -- masked entries become -inf before the row max.
        mask = np.fromiter(action_mask, dtype=np.int32, count=len(action_mask))
        assert mask.size == self.action_size
        if not deterministic and self._rng.random() < exploration_rate:
            cand = np.where(mask)[0]
        else:
            mq = np.where(mask, self.q_table[state], -np.inf)
            cand = np.where(mq == np.max(mq))[0]
        return self._rng.choice(cand)

    def _no_candidate(self):
        # reference :302, :348 return -1 here; the draw shim is told that this agent drew nothing
        skip = getattr(self._rng, "skip_choice", None)
        if skip is not None:
            skip()
        return -1

    # ------------------------------------------------------------------ selection, batched forms
    def choose_actions_iter(self, states, exploration_rate, *, deterministic=False, action_masks=None):
        # :350-400
        if action_masks is None:
            gen = (self.choose_action(s, exploration_rate, deterministic=deterministic) for s in states)
        else:
            gen = (
                self.choose_masked_action(s, m, exploration_rate, deterministic=deterministic)
                for s, m in zip(states, action_masks, strict=True)
            )
        return np.fromiter(gen, dtype=np.int32, count=len(states))

    def choose_actions_vec_iter(self, states, exploration_rate, *, deterministic=False, action_masks=None):
        # :472-522
        if action_masks is None:
            gen = (self.choose_action_vec(s, exploration_rate, deterministic=deterministic) for s in states)
        else:
            gen = (
                self.choose_masked_action_vec(s, m, exploration_rate, deterministic=deterministic)
                for s, m in zip(states, action_masks, strict=True)
            )
        return np.fromiter(gen, dtype=np.int32, count=len(states))

    def choose_actions_vec(self, states, exploration_rate, *, deterministic=False):
        # :524-579 -- one gather for the row maxima, batched explore flags + exploratory actions
        # from the NumPy generator, then one ``choice`` per greedy agent.
        rows = self.q_table[states]
        maxima = np.max(rows, axis=1, keepdims=True)
        if deterministic:
            gen = (self._rng.choice(np.where(r == m)[0]) for r, m in zip(rows, maxima, strict=False))
        else:
            explore = self._np_rng.random(states.size) < exploration_rate
            explore_actions = self._np_rng.integers(self.action_size, size=states.size)
            gen = (
                ea if ex else self._rng.choice(np.where(r == m)[0])
                for r, m, ex, ea in zip(rows, maxima, explore, explore_actions, strict=True)
            )
        return np.fromiter(gen, dtype=np.int32, count=states.size)

    def choose_masked_actions_vec(self, states, action_masks, exploration_rate, *, deterministic=False):
        # :581-642
        assert action_masks.shape == (states.size, self.action_size)
        mq = np.where(action_masks, self.q_table[states], -np.inf)
        maxima = np.max(mq, axis=1, keepdims=True)
        if deterministic:
            gen = (self._rng.choice(np.where(r == m)[0]) for r, m in zip(mq, maxima, strict=True))
        else:
            explore = self._np_rng.random(states.size) < exploration_rate
            gen = (
                self._rng.choice(np.where(k)[0]) if ex else self._rng.choice(np.where(r == m)[0])
                for r, m, k, ex in zip(mq, maxima, action_masks, explore, strict=True)
            )
        return np.fromiter(gen, dtype=np.int32, count=states.size)

    def choose_actions(self, states, exploration_rate, *, deterministic=False, action_masks=None):
        # :644-726 -- size-based dispatch.
        kw = {"exploration_rate": exploration_rate, "deterministic": deterministic}
        if deterministic:
            if self.action_size <= _DET_MAX_A_ITER:
                return self.choose_actions_iter(states, action_masks=action_masks, **kw)
            if self.action_size >= _DET_MIN_A_VEC_ITER and len(states) <= _DET_MAX_N_VEC_ITER:
                return self.choose_actions_vec_iter(states, action_masks=action_masks, **kw)
            if action_masks is not None:
                return self.choose_masked_actions_vec(states, action_masks, **kw)
            return self.choose_actions_vec(states, **kw)
        if action_masks is None:
            if len(states) < _NOMASK_MAX_N_ITER:
                return self.choose_actions_iter(states, **kw)
            if self.action_size > _NOMASK_MIN_A_VEC_ITER:
                return self.choose_actions_vec_iter(states, **kw)
            return self.choose_actions_vec(states, **kw)
        if self.action_size <= _MASK_MAX_A_ITER:
            return self.choose_actions_iter(states, action_masks=action_masks, **kw)
        return self.choose_actions_vec_iter(states, action_masks=action_masks, **kw)

    # ------------------------------------------------------------------ table accessors (:100-233)
    # kept as methods (not inlined) so that the per-transition call overhead matches the reference's
    def get_q_value(self, state, action):
        return self.q_table[state, action]

    def get_state_q_values(self, state):
        return self.q_table[state]

    def add_q_value(self, state, action, value):
        self.q_table[state, action] += value

    # ------------------------------------------------------------------ learning
    def single_learn(self, state, action, reward, next_state, terminated, lr, next_action_mask=None):
        # :728-768 -- target = r + gamma * max_valid Q[s'] (0 when terminated); in-place update.
        if next_action_mask is None:
            nxt = 0 if terminated else np.max(self.get_state_q_values(next_state))
        else:
            nxt = 0 if terminated else np.max(self.get_state_q_values(next_state)[np.where(next_action_mask)])
        target = reward + self.discount_factor * nxt
        prediction = self.get_q_value(state, action)
        self.add_q_value(state, action, lr * (target - prediction))

    def learn_iter(self, states, actions, rewards, next_states, terminated, lr, next_action_masks=None):
        # :770-817 -- strictly sequential over agents (agent i sees the writes of agents < i).
        if next_action_masks is None:
            for s, a, r, s2, te in zip(states, actions, rewards, next_states, terminated, strict=True):
                self.single_learn(s, a, r, s2, te, lr)
        else:
            for s, a, r, s2, te, m in zip(states, actions, rewards, next_states, terminated,
                                          next_action_masks, strict=True):
                self.single_learn(s, a, r, s2, te, lr, m)

    def learn_vec(self, states, actions, rewards, next_states, terminated, lr, next_action_masks=None):
        # :819-891 + add_q_values :235-250 -- all reads precede all writes; duplicates accumulate.
        rows = self.q_table[next_states]
        if next_action_masks is not None:
            rows = np.where(next_action_masks, rows, -np.inf)
        targets = rewards + self.discount_factor * np.max(rows, axis=1) * (1 - terminated)
        np.add.at(self.q_table, (states, actions), lr * (targets - self.q_table[states, actions]))

    def learn(self, states, actions, rewards, next_states, terminated, lr, next_action_masks=None):
        # :893-934 -- the dispatcher is hard-wired to the sequential form.
        self.learn_iter(states, actions, rewards, next_states, terminated, lr, next_action_masks)


# ---------------------------------------------------------------------------- schedules
class OracleSchedule:
    """``schedules/*.py`` in one class: kind in {"constant", "linear", "exponential"}.

    exponential (exponential_schedule.py:31):  v <- max(v * decay**steps, min_value)
    linear      (linear_schedule.py:31):       v <- v + steps * decay     (no clamp)
    constant    (constant_schedule.py:12):     no-op
    """

    def __init__(self, kind, value, min_value=None, decay=None):
        self.kind, self.value, self.min_value, self.decay = kind, value, min_value, decay

    def get_value(self):
        return self.value

    def update(self, steps):
        if self.kind == "exponential":
            self.value = max(self.value * (self.decay**steps), self.min_value)
        elif self.kind == "linear":
            self.value = self.value + steps * self.decay


# ---------------------------------------------------------------------------- runtime
def _n_agents(states):
    return len(states["observation"]) if isinstance(states, dict) else len(states)


class OracleRuntime:
    """Restates ``BaseRuntime`` + ``SingleThreadQLearning``
    (algorithms/runtime/base_runtime.py:184-384, single_thread_runtime.py:28-76)."""

    def __init__(self, algorithm, lr_schedule, exploration_rate_schedule, learn_mode="iter"):
        self.algorithm = algorithm
        self.lr_schedule = lr_schedule
        self.exploration_rate_schedule = exploration_rate_schedule
        self.learn_mode = learn_mode
        self.step_counter = 0  # vector steps taken; indexes the draw protocol
        self.trace = None  # optional list collecting (actions, eps, lr) per step

    def _begin_draws(self, n, eps, deterministic=False):
        d = self.algorithm._rng
        if isinstance(d, InjectedDraws):
            d.begin(self.step_counter, n, eps, deterministic=deterministic)

    def _choose_actions(self, states):
        # base_runtime.py:265-291
        eps = self.exploration_rate_schedule.get_value()
        self._begin_draws(_n_agents(states), eps)
        if isinstance(states, dict):
            return self.algorithm.choose_actions(
                states=states["observation"], action_masks=states["action_mask"], exploration_rate=eps
            )
        return self.algorithm.choose_actions(states, exploration_rate=eps)

    def _learn(self, states, actions, rewards, next_states, terminateds):
        # base_runtime.py:224-263 -- lr is read before the update, then both schedules advance by
        # the number of agents.
        fn = self.algorithm.learn if self.learn_mode == "iter" else self.algorithm.learn_vec
        lr = self.lr_schedule.get_value()
        if isinstance(next_states, dict):
            fn(states["observation"], actions, rewards, next_states["observation"], terminateds, lr,
               next_states["action_mask"])
            n = len(states["observation"])
        else:
            fn(states, actions, rewards, next_states, terminateds, lr)
            n = len(states)
        self.lr_schedule.update(n)
        self.exploration_rate_schedule.update(n)

    def run_single_step(self, env, states, agent_rewards, reward_history):
        # base_runtime.py:184-222
        actions = self._choose_actions(states)
        if self.trace is not None:
            self.trace.append(
                (actions.copy(), self.exploration_rate_schedule.get_value(), self.lr_schedule.get_value())
            )
        next_states, rewards, terminateds, truncateds, infos = env.step(actions)
        agent_rewards += rewards
        self._learn(states, actions, rewards, next_states, terminateds)
        self.step_counter += 1
        for i, (te, tr) in enumerate(zip(terminateds, truncateds, strict=True)):
            if te or tr:
                reward_history.append(agent_rewards[i])
                agent_rewards[i] = 0
        return next_states, infos

    def run_steps(self, steps, env, curr_state_dict=None):
        # single_thread_runtime.py:28-76 (the ZeroDivisionError on an empty history is kept).
        history = []
        if curr_state_dict is None:
            states, infos = env.reset()
            agent_rewards = np.zeros(_n_agents(states), dtype=np.float32)
        else:
            states, infos, agent_rewards = (curr_state_dict[k] for k in ("states", "infos", "rewards"))
        for _ in range(steps):
            states, infos = self.run_single_step(env, states, agent_rewards, history)
        return (
            sum(history) / len(history),
            history,
            env,
            {"states": states, "infos": infos, "rewards": agent_rewards, "episode_rewards": history},
        )

    def _greedy(self, states):
        self._begin_draws(_n_agents(states), 0.0, deterministic=True)
        if isinstance(states, dict):
            return self.algorithm.choose_actions(
                states=states["observation"], action_masks=states["action_mask"],
                exploration_rate=0.0, deterministic=True,
            )
        return self.algorithm.choose_actions(states, exploration_rate=0.0, deterministic=True)

    def evaluate_steps(self, env, steps):
        # base_runtime.py:293-336 -- greedy, reset(seed=42), ``steps // n`` vector steps.
        states, _ = env.reset(seed=42)
        n = _n_agents(states)
        acc, history = np.zeros(n, dtype=np.float32), []
        for _ in range(0, steps, n):
            states, rewards, te, tr, _ = env.step(self._greedy(states))
            self.step_counter += 1
            acc += rewards
            for i in range(n):
                if te[i] or tr[i]:
                    history.append(acc[i])
                    acc[i] = 0
        return sum(history), history

    def evaluate_episodes(self, env, episodes):
        # base_runtime.py:338-384 -- runs until ``episodes`` episodes have *ended* (the last vector
        # step may overshoot: every agent finishing in it is still recorded).
        states, _ = env.reset(seed=42)
        n = _n_agents(states)
        acc, history, done = np.zeros(n, dtype=np.float32), [], 0
        while done < episodes:
            states, rewards, te, tr, _ = env.step(self._greedy(states))
            self.step_counter += 1
            acc += rewards
            for i in range(n):
                if te[i] or tr[i]:
                    done += 1
                    history.append(acc[i])
                    acc[i] = 0
        return sum(history), history
