"""``BaseRuntime``: the driver contract of the reference
(``dist_classicrl/algorithms/runtime/base_runtime.py:23-384``) on top of the HIP engine.

``train`` / ``run_single_step`` / ``evaluate_*`` keep the reference's signatures, return values and
quirks (validation runs once even when "disabled", :156-169; ``train`` never feeds ``state_dict``
back into ``curr_state_dict``, :156-161).  The generic methods here drive *any* environment object
from the host, one vector step at a time, with selection and learning on the GPU; the fused,
device-resident loop lives in :mod:`.gpu_rollout_runtime`.
"""

from __future__ import annotations

import logging
from abc import ABC, abstractmethod

import numpy as np

logger = logging.getLogger(__name__)


def _count_agents(states) -> int:
    return len(states["observation"]) if isinstance(states, dict) else len(states)


class BaseRuntime(ABC):
    def __init__(self, algorithm, lr_schedule, exploration_rate_schedule) -> None:
        self.algorithm = algorithm
        self.lr_schedule = lr_schedule
        self.exploration_rate_schedule = exploration_rate_schedule

    @abstractmethod
    def init_training(self) -> None: ...

    @abstractmethod
    def run_steps(self, steps, env, curr_state_dict): ...

    @abstractmethod
    def close_training(self) -> None: ...

    # ------------------------------------------------------------------ train (:99-182)
    def train(self, env, steps, val_env, val_every_n_steps, val_steps=None, val_episodes=None,
              curr_state_dict=None):
        assert (val_steps is None) ^ (val_episodes is None), (
            "Exactly one of val_steps or val_episodes must be specified."
        )
        self.init_training()
        reward_history, val_reward_history = [], []
        state_dict = None
        for step in range(0, steps, val_every_n_steps):
            _, episode_rewards, env, state_dict = self.run_steps(
                steps=min(val_every_n_steps, steps - step), env=env, curr_state_dict=curr_state_dict
            )
            reward_history.extend(episode_rewards)
            if val_steps is not None:
                total, _per_agent = self.evaluate_steps(val_env, val_steps)
            else:
                total, _per_agent = self.evaluate_episodes(val_env, val_episodes)
            val_reward_history.append(total)
            logger.debug("Step %d, Eval total rewards: %s", step + 1, total)
        self.close_training()
        return reward_history, val_reward_history, env, state_dict

    # ------------------------------------------------------------------ one vector step (:184-291)
    def _choose_actions(self, states):
        eps = self.exploration_rate_schedule.get_value()
        if isinstance(states, dict):
            return self.algorithm.choose_actions(
                states=states["observation"], action_masks=states["action_mask"], exploration_rate=eps
            )
        return self.algorithm.choose_actions(states, exploration_rate=eps)

    def _learn(self, states, actions, rewards, next_states, terminateds) -> None:
        lr = self.lr_schedule.get_value()  # read before the schedules advance (:244, :259)
        if isinstance(next_states, dict):
            assert isinstance(states, dict)
            self.algorithm.learn(states["observation"], actions, rewards, next_states["observation"],
                                 terminateds, lr, next_states["action_mask"])
        else:
            assert not isinstance(states, dict)
            self.algorithm.learn(states, actions, rewards, next_states, terminateds, lr)
        n_updates = _count_agents(states)
        self.lr_schedule.update(n_updates)
        self.exploration_rate_schedule.update(n_updates)

    def run_single_step(self, env, states, agent_rewards, reward_history):
        actions = self._choose_actions(states)
        next_states, rewards, terminateds, truncateds, infos = env.step(actions)
        agent_rewards += rewards
        self._learn(states, actions, rewards, next_states, terminateds)
        for i in np.flatnonzero(np.logical_or(terminateds, truncateds)):
            reward_history.append(agent_rewards[i])
            agent_rewards[i] = 0
        return next_states, infos

    # ------------------------------------------------------------------ evaluation (:293-384)
    def _greedy_actions(self, states):
        if isinstance(states, dict):
            return self.algorithm.choose_actions(
                states=states["observation"], action_masks=states["action_mask"],
                exploration_rate=0.0, deterministic=True,
            )
        return self.algorithm.choose_actions(states, exploration_rate=0.0, deterministic=True)

    def _prepare_env(self, env):
        return env

    def evaluate_steps(self, env, steps):
        """Greedy rollout of ``steps // n_agents`` vector steps from ``reset(seed=42)``; returns
        (sum of finished-episode returns, list of them)."""
        env = self._prepare_env(env)
        states, _ = env.reset(seed=42)
        n = _count_agents(states)
        acc = np.zeros(n, dtype=np.float32)
        history = []
        for _ in range(0, steps, n):
            states, rewards, terminateds, truncateds, _infos = env.step(self._greedy_actions(states))
            acc += rewards
            for i in np.flatnonzero(np.logical_or(terminateds, truncateds)):
                history.append(acc[i])
                acc[i] = 0
        return sum(history), history

    def evaluate_episodes(self, env, episodes):
        """Greedy rollout until ``episodes`` episodes have ended (every episode ending in the last
        vector step is still recorded, as in the reference)."""
        env = self._prepare_env(env)
        states, _ = env.reset(seed=42)
        n = _count_agents(states)
        acc = np.zeros(n, dtype=np.float32)
        history = []
        while len(history) < episodes:
            states, rewards, terminateds, truncateds, _infos = env.step(self._greedy_actions(states))
            acc += rewards
            for i in np.flatnonzero(np.logical_or(terminateds, truncateds)):
                history.append(acc[i])
                acc[i] = 0
        return sum(history), history
