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


C_TTT = 0x7F4A7C15
LINES = (0b000000111, 0b000111000, 0b111000000, 0b001001001, 0b010010010, 0b100100100,
         0b100010001, 0b001010100)  # bit i = board cell i (row-major)
POW3 = 3 ** np.arange(8, -1, -1)  # cell 0 is the most significant base-3 digit (utils.py:12-29)


def ttt_winner(mask1: int, mask2: int):
    """``_check_winner`` (environments/tiktaktoe_mod.py:216-237) on occupancy bit masks."""
    for line in LINES:
        if mask1 & line == line:
            return 1
        if mask2 & line == line:
            return 2
    return None


def ttt_encode(mask1: int, mask2: int) -> int:
    """Flattened board (``_get_obs`` :199-214) through the mixed-radix encoder
    (``utils.py:32-48`` with ``compute_radix([3]*9)`` = [6561, ..., 3, 1])."""
    return int(sum((((mask1 >> i) & 1) + 2 * ((mask2 >> i) & 1)) * POW3[i] for i in range(9)))


class TicTacToeVecEnv(_VecEnvBase):
    """n copies of the reference's TicTacToe (environments/tiktaktoe_mod.py:67-237) behind the
    Flatten-MultiDiscrete wrapper: observation = base-3 board id (S = 19683), 9 actions, action mask
    = empty cells; the machine opponent plays uniformly random legal moves; reward +1 / -1 / 0;
    never truncates.  Only the *source* of randomness differs from the reference (which uses a NumPy
    Generator per env): three 32-bit words per (agent, vector step),

        h0 = mix32(mix32(agent ^ seed ^ C_TTT) + step_lo * 0x9E3779B9 + step_hi)
        machine reply = mulhi(h0, n_empty)-th empty cell
        on reset: h1 = mix32(h0 ^ 0x68E31DA4) -> agent starts iff h1 & 1 == 0
                  h2 = mix32(h0 ^ 0xB5297A4D) -> machine's opening cell = mulhi(h2, 9)

    State per agent: cells holding mark 1 / mark 2 (two 9-bit masks) and the agent's mark."""

    masked = True

    def __init__(self, num_agents, seed=1, agent_offset=0):
        self.num_agents = int(num_agents)
        self.state_size = 19683
        self.action_size = 9
        self.seed = int(seed) & 0xFFFFFFFF
        self.agent_ids = np.arange(agent_offset, agent_offset + self.num_agents, dtype=np.uint32)
        self.m1 = np.zeros(self.num_agents, dtype=np.int64)
        self.m2 = np.zeros(self.num_agents, dtype=np.int64)
        self.agent_mark = np.ones(self.num_agents, dtype=np.int64)
        self.step_index = 0  # vector step of the draw protocol the next step() belongs to

    def _words(self, step):
        inner = mix32(self.agent_ids ^ np.uint32(self.seed ^ C_TTT)).astype(np.uint64)
        mixed = (inner + np.uint64(step & 0xFFFFFFFF) * np.uint64(0x9E3779B9) + np.uint64(step >> 32))
        h0 = mix32(mixed & np.uint64(0xFFFFFFFF))
        return h0, mix32(h0 ^ np.uint32(0x68E31DA4)), mix32(h0 ^ np.uint32(0xB5297A4D))

    def _begin_episode(self, i, h1, h2):
        self.m1[i] = self.m2[i] = 0
        if (int(h1) & 1) == 0:  # agent starts: it plays mark 1 (:93-96)
            self.agent_mark[i] = 1
        else:  # machine starts with mark 1 on a uniformly random cell (:97-101)
            self.agent_mark[i] = 2
            self.m1[i] = 1 << int(mulhi32(h2, 9))

    def _obs(self):
        obs = np.array([ttt_encode(int(a), int(b)) for a, b in zip(self.m1, self.m2)], dtype=np.int32)
        return {"observation": obs, "action_mask": self.action_masks(obs)}

    def action_masks(self, obs):
        digits = (np.asarray(obs, dtype=np.int64)[:, None] // POW3[None, :]) % 3
        return (digits == 0).astype(np.int8)

    def reset(self, seed=None, options=None):  # noqa: ARG002
        if seed is not None:
            self.seed = int(seed) & 0xFFFFFFFF
        _, h1, h2 = self._words(0xFFFFFFFFFFFFFFFF)  # a step index no rollout reaches
        for i in range(self.num_agents):
            self._begin_episode(i, h1[i], h2[i])
        return self._obs(), [{} for _ in range(self.num_agents)]

    def step(self, actions):
        h0, h1, h2 = self._words(self.step_index)
        self.step_index += 1
        n = self.num_agents
        rewards = np.zeros(n, dtype=np.float32)
        terminated = np.zeros(n, dtype=bool)
        for i in range(n):
            mine, theirs = (self.m1, self.m2) if self.agent_mark[i] == 1 else (self.m2, self.m1)
            mine[i] |= 1 << int(actions[i])  # agent's move (:155-158)
            full = lambda: (self.m1[i] | self.m2[i]) == 0x1FF  # noqa: E731
            if ttt_winner(int(self.m1[i]), int(self.m2[i])) == self.agent_mark[i]:
                rewards[i], terminated[i] = 1.0, True
            elif full():
                terminated[i] = True
            else:  # machine's reply: uniformly random empty cell (:160-166, 173-190)
                empty = [c for c in range(9) if not ((self.m1[i] | self.m2[i]) >> c) & 1]
                theirs[i] |= 1 << empty[int(mulhi32(h0[i], len(empty)))]
                if ttt_winner(int(self.m1[i]), int(self.m2[i])) is not None:
                    rewards[i], terminated[i] = -1.0, True
                elif full():
                    terminated[i] = True
            if terminated[i]:  # SAME_STEP autoreset
                self._begin_episode(i, h1[i], h2[i])
        truncated = np.zeros(n, dtype=bool)
        return self._obs(), rewards, terminated, truncated, [{}] * n
