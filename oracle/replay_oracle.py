"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's experience-replay ring
(``algorithms/buffers/experience_replay.py:13-120``), pinned by ``tests/golden/replay.npz`` (generated
from the real class).  The product never imports this module."""

from __future__ import annotations

import numpy as np


class OracleReplay:
    def __init__(self, capacity: int, seed: int) -> None:  # :56-66
        self.capacity = capacity
        self.state_buffer = np.zeros(capacity, dtype=np.int64)
        self.action_buffer = np.zeros(capacity, dtype=np.int64)
        self.reward_buffer = np.zeros(capacity, dtype=np.float64)
        self.next_state_buffer = np.zeros(capacity, dtype=np.int64)
        self.done_buffer = np.zeros(capacity, dtype=bool)
        self.position = 0
        self.full = False
        self.rng = np.random.default_rng(seed)

    def push(self, experience) -> None:  # :68-86
        s, a, r, n, d = experience
        p = self.position
        self.state_buffer[p], self.action_buffer[p], self.reward_buffer[p] = s, a, r
        self.next_state_buffer[p], self.done_buffer[p] = n, d
        self.position = (p + 1) % self.capacity
        self.full = self.full or self.position == 0

    def indices(self, batch_size: int) -> np.ndarray:  # :103-105
        return self.rng.choice(self.capacity if self.full else self.position, batch_size, replace=False)

    def sample_arrays(self, batch_size: int):
        i = self.indices(batch_size)
        return (self.state_buffer[i], self.action_buffer[i], self.reward_buffer[i], self.next_state_buffer[i],
                self.done_buffer[i])

    def sample(self, batch_size: int):  # :88-109 (scalar conversions: batch_size == 1 only)
        s, a, r, n, d = self.sample_arrays(batch_size)
        if s.size != 1:
            msg = "only length-1 arrays can be converted to Python scalars"
            raise TypeError(msg)
        return (int(s[0]), int(a[0]), float(r[0]), int(n[0]), bool(d[0]))

    def __len__(self) -> int:  # :111-120
        return self.capacity if self.full else self.position
