"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the engine's batched tabular environments.

The reference ships no environment at the BASELINE shapes; these integer-only
environments are *defined by this build* so that CPU and GPU agree bit for bit.  They follow the
reference's multi-agent contract (``environments/custom_env.py:31-69``) as realised by
``SyncVectorEnv(..., autoreset_mode=SAME_STEP)`` (``benchmarks/throughput_benchmark.py:109-123``):

    reset(seed=None, options=None) -> (obs | {"observation", "action_mask"}, infos)
    step(actions int32[n]) -> (obs, rewards float32[n], terminated bool[n], truncated bool[n], infos)

On termination the returned observation is already the first observation of the next episode.
``truncated`` is always False (no reference environment truncates).
"""

from __future__ import annotations

import numpy as np

from .draws import mix32, mulhi32

C_REWARD = 0x9E3779B9
C_TERM = 0x85EBCA6B
C_MASK = 0xA511E9B3
C_HOLE = 0x1B873593
C_START = 0x2545F491


class _VecEnvBase:
    num_agents: int
    state_size: int
    action_size: int
    masked = False

    def _wrap(self, obs):
        if self.masked:
            return {"observation": obs.copy(), "action_mask": self.action_masks(obs)}
        return obs.copy()

    def __len__(self) -> int:  # some callers use len(env) for the agent count
        return self.num_agents


class HashTabularEnv(_VecEnvBase):
    """Deterministic hashed MDP: ``s' = mix32((s*A + a) ^ seed) * S >> 32``.

    reward      = (mix32(s' ^ C_REWARD ^ seed) >> 8) * 2**-24                (float32 exact)
    terminated  = (mix32(s' ^ C_TERM ^ seed) & 0xff) < p_term_256
    start state = mix32(mix32(agent ^ seed ^ C_START) + episode * 0x9E3779B9) * S >> 32
    mask word k = mix32((s * n_words + k) ^ seed ^ C_MASK); action 0 always valid
    """

    def __init__(self, num_agents, state_size, action_size, seed=1, p_term_256=13, masked=False,
                 agent_offset=0):
        self.num_agents = int(num_agents)
        self.state_size = int(state_size)
        self.action_size = int(action_size)
        self.seed = int(seed) & 0xFFFFFFFF
        self.p_term_256 = int(p_term_256)
        self.masked = bool(masked)
        self.agent_ids = np.arange(agent_offset, agent_offset + self.num_agents, dtype=np.uint32)
        self.obs = np.zeros(self.num_agents, dtype=np.int32)
        self.episode = np.zeros(self.num_agents, dtype=np.uint32)

    def _start_states(self, episode):
        inner = mix32(self.agent_ids ^ np.uint32(self.seed ^ C_START)).astype(np.uint64)
        h = mix32((inner + np.asarray(episode, dtype=np.uint64) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF))
        return mulhi32(h, self.state_size).astype(np.int32)

    def action_masks(self, obs):
        n_words = (self.action_size + 31) // 32
        obs = np.asarray(obs, dtype=np.uint64)
        cols = np.arange(self.action_size)
        words = mix32(
            (obs[:, None] * np.uint64(n_words) + np.arange(n_words, dtype=np.uint64)[None, :])
            ^ np.uint64(self.seed ^ C_MASK)
        )
        bits = (words[:, cols // 32] >> (cols % 32).astype(np.uint32)) & np.uint32(1)
        bits[:, 0] = 1
        return bits.astype(np.int8)

    def reset(self, seed=None, options=None):  # noqa: ARG002
        if seed is not None:
            self.seed = int(seed) & 0xFFFFFFFF
        self.episode[:] = 0
        self.obs = self._start_states(self.episode)
        return self._wrap(self.obs), [{} for _ in range(self.num_agents)]

    def step(self, actions):
        a = np.asarray(actions).astype(np.uint64)
        key = (self.obs.astype(np.uint64) * np.uint64(self.action_size) + a) & np.uint64(0xFFFFFFFF)
        nxt = mulhi32(mix32(key ^ np.uint64(self.seed)), self.state_size)
        rewards = (mix32(nxt ^ np.uint32(self.seed ^ C_REWARD)) >> np.uint32(8)).astype(
            np.float32
        ) * np.float32(2.0**-24)
        terminated = (mix32(nxt ^ np.uint32(self.seed ^ C_TERM)) & np.uint32(0xFF)) < self.p_term_256
        self.episode = self.episode + terminated.astype(np.uint32)
        self.obs = np.where(terminated, self._start_states(self.episode), nxt.astype(np.int32))
        self.obs = self.obs.astype(np.int32)
        truncated = np.zeros(self.num_agents, dtype=bool)
        return self._wrap(self.obs), rewards, terminated, truncated, [{}] * self.num_agents


class GridLakeEnv(_VecEnvBase):
    """FrozenLake-style ``side x side`` grid, deterministic moves (0=left 1=down 2=right 3=up).

    Start = cell 0, goal = last cell (reward 1, terminates), hash-placed holes (reward 0,
    terminate): ``hole(p) = p not in {0, goal} and mix32(p ^ seed ^ C_HOLE) % 5 == 0``.
    Every agent restarts at cell 0, so same-state collisions between agents are the norm.
    """

    def __init__(self, num_agents, side=10, seed=1):
        self.num_agents = int(num_agents)
        self.side = int(side)
        self.state_size = self.side * self.side
        self.action_size = 4
        self.seed = int(seed) & 0xFFFFFFFF
        self.obs = np.zeros(self.num_agents, dtype=np.int32)

    def holes(self):
        p = np.arange(self.state_size, dtype=np.uint32)
        h = (mix32(p ^ np.uint32(self.seed ^ C_HOLE)) % np.uint32(5)) == 0
        h[0] = False
        h[-1] = False
        return h

    def reset(self, seed=None, options=None):  # noqa: ARG002
        if seed is not None:
            self.seed = int(seed) & 0xFFFFFFFF
        self.obs[:] = 0
        return self.obs.copy(), [{} for _ in range(self.num_agents)]

    def step(self, actions):
        a = np.asarray(actions).astype(np.int64)
        row, col = np.divmod(self.obs.astype(np.int64), self.side)
        col = np.clip(col + (a == 2) - (a == 0), 0, self.side - 1)
        row = np.clip(row + (a == 1) - (a == 3), 0, self.side - 1)
        nxt = (row * self.side + col).astype(np.int32)
        goal = nxt == self.state_size - 1
        terminated = goal | self.holes()[nxt]
        rewards = goal.astype(np.float32)
        self.obs = np.where(terminated, 0, nxt).astype(np.int32)
        truncated = np.zeros(self.num_agents, dtype=bool)
        return self.obs.copy(), rewards, terminated, truncated, [{}] * self.num_agents


class RiggedBanditVecEnv(_VecEnvBase):
    """n independent copies of the reference's known-answer environment
    (``environments/rigged_two_armed_bandit.py:55-80``): one state, ``reward = action``, terminates
    every ``episode_len`` steps and restarts its own counter."""

    def __init__(self, num_agents, episode_len=10):
        self.num_agents = int(num_agents)
        self.state_size = 1
        self.action_size = 2
        self.episode_len = int(episode_len)
        self.t = np.zeros(self.num_agents, dtype=np.int32)
        self.obs = np.zeros(self.num_agents, dtype=np.int32)

    def reset(self, seed=None, options=None):  # noqa: ARG002
        self.t[:] = 0
        return self.obs.copy(), [{} for _ in range(self.num_agents)]

    def step(self, actions):
        rewards = np.asarray(actions).astype(np.float32)
        self.t += 1
        terminated = self.t >= self.episode_len
        self.t[terminated] = 0
        truncated = np.zeros(self.num_agents, dtype=bool)
        return self.obs.copy(), rewards, terminated, truncated, [{}] * self.num_agents
