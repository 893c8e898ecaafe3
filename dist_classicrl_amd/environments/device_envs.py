"""Batched tabular environments whose state lives on the GPU next to the Q-table.

They follow the reference's vector-env contract as realised by
``SyncVectorEnv(autoreset_mode=SAME_STEP)`` (``benchmarks/throughput_benchmark.py:109-123``):

    reset(seed=None, options=None) -> (obs | {"observation", "action_mask"}, infos)
    step(actions) -> (obs, rewards float32[n], terminated bool[n], truncated bool[n], infos)

so the generic host-driven loop of ``BaseRuntime`` works with them, but their point is the fused
path: ``GpuRolloutQLearning.run_steps`` keeps select -> env.step -> learn on the device
(``qe_rollout``).  An environment must be bound to the algorithm whose GPU it shares
(:meth:`DeviceVecEnv.bind`; the runtimes do that themselves).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from dist_classicrl_amd import _lib


class DeviceVecEnv:
    kind = -1
    masked = False

    def __init__(self, num_agents: int, state_size: int, action_size: int, params: _lib.EnvParams):
        self.num_agents = int(num_agents)
        self.state_size = int(state_size)
        self.action_size = int(action_size)
        self._params = params
        self._h = C.c_void_p()
        self._algo = None
        self._lib = None
        self._resident = None  # (states array, rewards array) of the state dict that mirrors the device
        self._chunk_limits = {}

    def __len__(self) -> int:
        return self.num_agents

    def __del__(self):
        self.close()

    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h is not None and h.value and self._lib is not None:
            self._lib.qe_env_destroy(h)
            self._h = C.c_void_p()

    @property
    def handle(self):
        return self._h

    def bind(self, algorithm) -> "DeviceVecEnv":
        """Create the device state on ``algorithm``'s GPU/stream (idempotent per algorithm)."""
        if self._algo is algorithm and self._h.value:
            return self
        if (algorithm.state_size, algorithm.action_size) != (self.state_size, self.action_size):
            msg = (
                f"environment is {self.state_size} states x {self.action_size} actions but the "
                f"algorithm's table is {algorithm.state_size} x {algorithm.action_size}"
            )
            raise ValueError(msg)
        self.close()
        self._resident = None
        self._chunk_limits = {}
        self._lib = _lib.load()
        _lib.check(self._lib.qe_env_create(C.byref(self._h), algorithm.handle, self.num_agents,
                                           C.byref(self._params)))
        self._algo = algorithm
        return self

    def _need(self):
        if not self._h.value:
            msg = "device environment is not bound: call env.bind(algorithm) first"
            raise RuntimeError(msg)

    def _wrap(self, obs, masks):
        if self.masked:
            return {"observation": obs, "action_mask": masks.view(np.int8)}
        return obs

    def observe(self):
        """(observations, running per-agent episode returns) as host arrays."""
        self._need()
        obs = np.empty(self.num_agents, dtype=np.int32)
        acc = np.empty(self.num_agents, dtype=np.float32)
        masks = np.empty((self.num_agents, self.action_size), dtype=np.uint8) if self.masked else None
        _lib.check(self._lib.qe_env_observe(self._h, _lib.ptr(obs, C.c_int32), _lib.ptr(masks, C.c_uint8),
                                            _lib.ptr(acc, C.c_float)))
        return self._wrap(obs, masks), acc

    def aux(self) -> np.ndarray:
        """Environment-internal per-agent state (episode counters, board marks): what an exact resume
        needs besides observations and running returns."""
        self._need()
        out = np.empty(self.num_agents, dtype=np.uint32)
        _lib.check(self._lib.qe_env_aux(self._h, _lib.ptr(out, C.c_uint32)))
        return out

    def state_dict(self) -> dict:
        """The resume dict of ``SingleThreadQLearning.run_steps`` (single_thread_runtime.py:70-75) for
        the current device state, plus ``"aux"``.  Its arrays are read-only: as long as the caller hands
        this very dict (or these very arrays) back, :meth:`is_resident` recognises it and nothing is
        copied to the device."""
        states, rewards = self.observe()
        aux = self.aux()
        obs = states["observation"] if isinstance(states, dict) else states
        for arr in (obs, rewards, aux):
            arr.flags.writeable = False
        self._resident = (obs, rewards)
        return {"states": states, "infos": [{}] * self.num_agents, "rewards": rewards, "aux": aux}

    def adopt_state(self, obs, rewards, aux) -> dict:
        """:meth:`state_dict` from arrays the engine has just filled (no further device round trip)."""
        for arr in (obs, rewards, aux):
            arr.flags.writeable = False
        self._resident = (obs, rewards)
        return {"states": obs, "infos": [{}] * self.num_agents, "rewards": rewards, "aux": aux}

    def chunk_limit(self, learn: bool) -> int:
        """Vector steps one launch may take on this environment (``qe_rollout_chunk_limit``; cached)."""
        key = bool(learn)
        if key not in self._chunk_limits:
            self._chunk_limits[key] = max(1, int(self._lib.qe_rollout_chunk_limit(self._algo.handle, self._h, 1 if learn else 0)))
        return self._chunk_limits[key]

    def is_resident(self, state_dict) -> bool:
        """True if ``state_dict`` is the (unmodified) one :meth:`state_dict` produced last and the device
        state has not been touched since."""
        if self._resident is None:
            return False
        states = state_dict.get("states")
        obs = states.get("observation") if isinstance(states, dict) else states
        rewards = state_dict.get("rewards")
        return (obs is self._resident[0] and rewards is self._resident[1]
                and not obs.flags.writeable and not rewards.flags.writeable)

    def restore(self, obs=None, agent_rewards=None, aux=None) -> None:
        self._need()
        self._resident = None
        if isinstance(obs, dict):
            obs = obs["observation"]
        o = None if obs is None else _lib.as_i32(obs)
        r = None if agent_rewards is None else np.ascontiguousarray(agent_rewards, dtype=np.float32)
        x = None if aux is None else np.ascontiguousarray(aux, dtype=np.uint32)
        for arr in (o, r, x):
            if arr is not None and arr.size != self.num_agents:
                msg = f"expected {self.num_agents} entries, got {arr.size}"
                raise ValueError(msg)
        _lib.check(self._lib.qe_env_restore(self._h, _lib.ptr(o, C.c_int32), _lib.ptr(x, C.c_uint32),
                                            _lib.ptr(r, C.c_float)))

    def reset_device(self, seed=None) -> None:
        """``reset`` without fetching the observations (the fused rollout does not need them)."""
        self._need()
        self._resident = None
        _lib.check(self._lib.qe_env_reset(self._h, 0 if seed is None else 1,
                                          0 if seed is None else int(seed) & 0xFFFFFFFF))

    def reset(self, seed=None, options=None):  # noqa: ARG002
        self.reset_device(seed)
        obs, _ = self.observe()
        return obs, [{}] * self.num_agents

    def step(self, actions):
        self._need()
        self._resident = None
        a = _lib.as_i32(actions).ravel()
        if a.size != self.num_agents:
            msg = f"expected {self.num_agents} actions, got {a.size}"
            raise ValueError(msg)
        n = self.num_agents
        obs = np.empty(n, dtype=np.int32)
        rewards = np.empty(n, dtype=np.float32)
        term = np.empty(n, dtype=np.uint8)
        masks = np.empty((n, self.action_size), dtype=np.uint8) if self.masked else None
        _lib.check(self._lib.qe_env_step(self._h, _lib.ptr(a, C.c_int32), _lib.ptr(obs, C.c_int32),
                                         _lib.ptr(rewards, C.c_float), _lib.ptr(term, C.c_uint8),
                                         _lib.ptr(masks, C.c_uint8)))
        return self._wrap(obs, masks), rewards, term.astype(bool), np.zeros(n, dtype=bool), [{}] * n


class HashTabularEnv(DeviceVecEnv):
    """Synthetic hashed MDP at the BASELINE shapes (definition: ``csrc/qe_envs.h`` / SURVEY 8d)."""

    kind = _lib.ENV_HASH

    def __init__(self, num_agents, state_size, action_size, seed=1, p_term_256=13, masked=False,
                 agent_offset=0):
        p = _lib.EnvParams(kind=self.kind, masked=int(bool(masked)), seed=int(seed) & 0xFFFFFFFF,
                           p_term_256=int(p_term_256), agent_offset=int(agent_offset))
        super().__init__(num_agents, state_size, action_size, p)
        self.masked = bool(masked)


class GridLakeEnv(DeviceVecEnv):
    """FrozenLake-style ``side x side`` grid with deterministic moves (BASELINE config 1)."""

    kind = _lib.ENV_GRID

    def __init__(self, num_agents, side=10, seed=1):
        p = _lib.EnvParams(kind=self.kind, seed=int(seed) & 0xFFFFFFFF, side=int(side))
        super().__init__(num_agents, int(side) * int(side), 4, p)


class RiggedTwoArmedBanditVecEnv(DeviceVecEnv):
    """``n`` copies of ``environments/rigged_two_armed_bandit.py:55-80`` (known-answer fixture)."""

    kind = _lib.ENV_BANDIT

    def __init__(self, num_agents, episode_len=10):
        p = _lib.EnvParams(kind=self.kind, episode_len=int(episode_len))
        super().__init__(num_agents, 1, 2, p)


class TicTacToeEnv(DeviceVecEnv):
    """``n`` games of the reference's TicTacToe against a uniformly random opponent
    (``environments/tiktaktoe_mod.py:67-237``) behind its Flatten-MultiDiscrete wrapper
    (``wrappers/flatten_multidiscrete_wrapper.py:106-161``): 19 683 states, 9 masked actions --
    the environment of every number the reference publishes (``docs/benchmarks.rst``)."""

    kind = _lib.ENV_TICTACTOE
    masked = True

    def __init__(self, num_agents, seed=1, agent_offset=0):
        p = _lib.EnvParams(kind=self.kind, masked=1, seed=int(seed) & 0xFFFFFFFF, agent_offset=int(agent_offset))
        super().__init__(num_agents, 19683, 9, p)
