"""Device-resident mirror of the reference's ``ExperienceReplay``
(``algorithms/buffers/experience_replay.py:13-120``; marked WIP upstream and used by no runtime).

Same constructor, attributes and methods.  The ring lives in HBM (``qe_replay_*`` in
``include/qlearn_engine.h``); the choice of WHICH entries to sample stays on the host with the same
``numpy.random.default_rng(seed).choice(len, batch_size, replace=False)`` call the reference makes
(:103-105), so with equal seeds both pick the same entries.

Kept quirks: ``sample`` converts its five result arrays with ``int()`` / ``float()`` / ``bool()``
(:104-109), which only works for ``batch_size == 1`` -- larger batches raise ``TypeError`` exactly like
the reference.  Added for actual use: :meth:`push_batch`, :meth:`sample_arrays` and :meth:`learn_from`
(gather + ``learn`` without leaving the device).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from dist_classicrl_amd import _lib


class ExperienceReplay:
    def __init__(self, capacity: int, seed: int, device: int = 0) -> None:
        self._lib = _lib.load()
        self.capacity = int(capacity)
        self._h = C.c_void_p()
        _lib.check(self._lib.qe_replay_create(C.byref(self._h), int(device), self.capacity))
        self.rng = np.random.default_rng(seed)

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.qe_replay_destroy(h)

    # ------------------------------------------------------------------ bookkeeping (:63-66, :111-120)
    @property
    def position(self) -> int:
        return int(self._lib.qe_replay_position(self._h))

    @property
    def full(self) -> bool:
        return bool(self._lib.qe_replay_full(self._h))

    def __len__(self) -> int:
        return int(self._lib.qe_replay_len(self._h))

    # ------------------------------------------------------------------ storage views (:58-62)
    def _all(self):
        return self._gather(np.arange(self.capacity, dtype=np.int64))

    @property
    def state_buffer(self) -> np.ndarray:
        return self._all()[0]

    @property
    def action_buffer(self) -> np.ndarray:
        return self._all()[1]

    @property
    def reward_buffer(self) -> np.ndarray:
        return self._all()[2]

    @property
    def next_state_buffer(self) -> np.ndarray:
        return self._all()[3]

    @property
    def done_buffer(self) -> np.ndarray:
        return self._all()[4]

    # ------------------------------------------------------------------ push (:68-86)
    def push(self, experience) -> None:
        state, action, reward, next_state, done = experience
        self.push_batch([state], [action], [reward], [next_state], [done])

    def push_batch(self, states, actions, rewards, next_states, dones) -> None:
        """``push`` for many experiences at once, in order."""
        s = np.ascontiguousarray(states, dtype=np.int64).ravel()
        a = np.ascontiguousarray(actions, dtype=np.int64).ravel()
        r = np.ascontiguousarray(rewards, dtype=np.float64).ravel()
        n = np.ascontiguousarray(next_states, dtype=np.int64).ravel()
        d = np.ascontiguousarray(np.asarray(dones).astype(bool), dtype=np.uint8).ravel()
        if not (a.size == r.size == n.size == d.size == s.size):
            msg = "experience arrays have different lengths"
            raise ValueError(msg)
        _lib.check(self._lib.qe_replay_push(self._h, _lib.ptr(s, C.c_int64), _lib.ptr(a, C.c_int64),
                                            _lib.ptr(r, C.c_double), _lib.ptr(n, C.c_int64),
                                            _lib.ptr(d, C.c_uint8), s.size))

    def attach(self, algorithm) -> None:
        """Wire the ring to ``algorithm``'s fused rollouts: every transition of every agent and vector step
        is pushed device to device, in (step, agent) order -- what a host loop calling :meth:`push` after
        every ``env.step`` would store."""
        _lib.check(self._lib.qe_replay_attach(algorithm.handle, self._h))

    def detach(self, algorithm) -> None:
        _lib.check(self._lib.qe_replay_attach(algorithm.handle, None))

    # ------------------------------------------------------------------ sample (:88-109)
    def _indices(self, batch_size: int) -> np.ndarray:
        return self.rng.choice(self.capacity if self.full else self.position, batch_size, replace=False)

    def _gather(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.int64).ravel()
        k = idx.size
        s, a, n = (np.empty(k, dtype=np.int64) for _ in range(3))
        r, d = np.empty(k, dtype=np.float64), np.empty(k, dtype=np.uint8)
        _lib.check(self._lib.qe_replay_gather(self._h, _lib.ptr(idx, C.c_int64), k, _lib.ptr(s, C.c_int64),
                                              _lib.ptr(a, C.c_int64), _lib.ptr(r, C.c_double),
                                              _lib.ptr(n, C.c_int64), _lib.ptr(d, C.c_uint8)))
        return s, a, r, n, d.astype(bool)

    def sample(self, batch_size: int):
        s, a, r, n, d = self._gather(self._indices(batch_size))
        if s.size != 1:  # upstream: int(array) -- "only length-1 arrays can be converted to Python scalars"
            msg = "only length-1 arrays can be converted to Python scalars"
            raise TypeError(msg)
        return (int(s[0]), int(a[0]), float(r[0]), int(n[0]), bool(d[0]))

    def sample_arrays(self, batch_size: int):
        """The sampled batch as arrays ``(states, actions, rewards, next_states, dones)``."""
        return self._gather(self._indices(batch_size))

    def learn_from(self, algorithm, batch_size: int, lr: float, mode: str = "iter") -> np.ndarray:
        """Sample ``batch_size`` experiences and apply ``algorithm.learn`` (``mode="iter"``) or
        ``learn_vec`` to them on the device; returns the sampled indices."""
        idx = np.ascontiguousarray(self._indices(batch_size), dtype=np.int64)
        code = _lib.LEARN_ITER if mode == "iter" else _lib.LEARN_VEC
        _lib.check(self._lib.qe_replay_learn(self._h, algorithm.handle, _lib.ptr(idx, C.c_int64), idx.size,
                                             float(lr), code))
        return idx
