"""TEST INFRASTRUCTURE ONLY -- the random-draw protocol shared by the oracle and the HIP engine.

The reference consumes two *sequential* generators (CPython ``random.Random`` and NumPy
``default_rng``) in a variant-dependent order (reference
``algorithms/base_algorithms/q_learning_optimal.py:287-300, 426-430, 464-470, 551-570``).  A GPU
cannot (and should not) replay a sequential stream, so the engine defines a *counter based* protocol
and parity is established the way the reference's own tests do it
(``tests/dist_classicrl/algorithms/runtime/test_q_learning_runtimes.py:17-45``): the draws are
injected into the reference through ``_rng`` / ``_np_rng`` shims.

Protocol (one Philox4x32-10 block per (agent, vector step)):

    key     = (seed & 0xffffffff, seed >> 32)
    counter = (agent_id, step & 0xffffffff, step >> 32, STREAM_POLICY)
    x0 -> exploration test :  explore  <=>  x0 * 2**-32 < epsilon  <=>  x0 < ceil(epsilon * 2**32)
    x1 -> exploratory pick :  k = mulhi32(x1, n_candidates)   (k-th valid action, ascending)
    x2 -> greedy tie pick  :  k = mulhi32(x2, n_ties)         (k-th tied action, ascending)
    x3 -> unused

Environment randomness uses STREAM_ENV with the environment's own seed.
"""

from __future__ import annotations

import math

import numpy as np

STREAM_POLICY = 0
STREAM_ENV = 1

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)
_SH32 = np.uint64(32)


def philox4x32(c0, c1, c2, c3, k0: int, k1: int, rounds: int = 10):
    """Vectorised Philox4x32 (Salmon et al., SC'11).  Inputs broadcast; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *(np.asarray(c, dtype=np.uint64) & _MASK32 for c in (c0, c1, c2, c3))
    )
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(rounds):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = ((p1 >> _SH32) ^ c1 ^ np.uint64(k0)) & _MASK32
        n1 = p1 & _MASK32
        n2 = ((p0 >> _SH32) ^ c3 ^ np.uint64(k1)) & _MASK32
        n3 = p0 & _MASK32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def mix32(x):
    """murmur3 fmix32 finaliser on uint32 arrays / ints (returns uint32 array)."""
    x = np.asarray(x, dtype=np.uint64) & _MASK32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _MASK32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _MASK32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def mulhi32(x, n):
    """floor(x * n / 2**32) for uint32 ``x`` and 0 <= n < 2**32: unbiased-enough range reduction."""
    return ((np.asarray(x, dtype=np.uint64) * np.uint64(n)) >> _SH32).astype(np.uint32)


def epsilon_threshold(eps: float) -> int:
    """Integer threshold T such that ``x0 * 2**-32 < eps`` (float64 compare) <=> ``x0 < T``.

    ``eps * 2**32`` is exact in float64 (power-of-two scaling), so the equivalence is exact.
    """
    if not (eps > 0.0):  # also catches NaN
        return 0
    if eps >= 1.0:
        return 1 << 32
    return min(int(math.ceil(eps * 4294967296.0)), 1 << 32)


def policy_draws(seed: int, agent_ids, step: int):
    """(x0, x1, x2) uint32 arrays for the given agents at vector step ``step``."""
    x0, x1, x2, _ = philox4x32(
        agent_ids, step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF, STREAM_POLICY, seed, seed >> 32
    )
    return x0, x1, x2


class InjectedDraws:
    """Stateful stand-in for BOTH ``algo._rng`` and ``algo._np_rng`` of the reference class.

    It hands the protocol's draws to whichever reference variant is running.  Call
    :meth:`begin` before every ``choose_actions*`` call.  ``epsilon`` must be the exploration rate
    passed to that call (the shim needs it to know which draw -- x1 or x2 -- a ``choice`` serves).
    """

    def __init__(self, seed: int, agent_ids=None) -> None:
        self.seed = int(seed)
        self.agent_ids = agent_ids
        self._x = None

    def begin(self, step: int, n: int, epsilon: float, *, deterministic: bool = False) -> None:
        ids = np.arange(n, dtype=np.uint32) if self.agent_ids is None else self.agent_ids[:n]
        self._x = policy_draws(self.seed, ids, step)
        self._thr = 0 if deterministic else epsilon_threshold(epsilon)
        self._n = n
        self._cur = -1  # agent served by the latest scalar uniform/random call
        self._batch = False  # True once the NumPy-style batched random(n) was used
        self._choice_calls = 0
        self._deterministic = deterministic
        # ``choose_actions_vec`` (unmasked) draws exploratory actions with ``integers`` and then
        # calls ``choice`` only for greedy agents; ``choose_masked_actions_vec`` calls ``choice``
        # for every agent.  ``integers`` being called is what tells the two apart.
        self._skip_explorers = False

    # ---- NumPy Generator surface (reference :551-552, :614) ------------------------------
    def random(self, size=None):
        if size is None:  # CPython ``random.Random.random()`` surface (reference :427, :464)
            self._cur += 1
            return float(self._x[0][self._cur]) * 2.0**-32
        assert size == self._n
        self._batch = True
        return self._x[0].astype(np.float64) * 2.0**-32

    def integers(self, high, size=None):
        assert size == self._n
        self._skip_explorers = True
        return mulhi32(self._x[1], int(high)).astype(np.int64)

    # ---- CPython Random surface (reference :287-300, :336-348) -----------------------------
    def uniform(self, a: float, b: float) -> float:
        self._cur += 1
        return a + (b - a) * (float(self._x[0][self._cur]) * 2.0**-32)

    def randint(self, a: int, b: int) -> int:
        return a + int(mulhi32(self._x[1][self._cur], b - a + 1))

    def _explores(self, i: int) -> bool:
        return int(self._x[0][i]) < self._thr

    def skip_choice(self) -> None:
        """An agent of a list variant had no candidate and returned -1 WITHOUT calling ``choice``
        (reference :302, :348).  In deterministic calls the shim counts ``choice`` calls to know which
        agent is being served, so the oracle tells it about the agent that made none.  (The real
        reference cannot; the golden cases therefore contain no such agent.)"""
        if self._batch or self._deterministic or self._cur < 0:
            self._choice_calls += 1

    def choice(self, seq):
        n = len(seq)
        if n == 0:
            msg = "Cannot choose from an empty sequence"
            raise IndexError(msg)
        if self._batch or self._deterministic or self._cur < 0:
            # batched variants / deterministic calls: one ``choice`` per agent that needs one, in
            # agent order.  In ``choose_actions_vec`` only non-exploring agents call it.
            i = self._choice_calls
            if self._batch and self._skip_explorers:
                while self._explores(i):
                    i += 1
            self._choice_calls = i + 1
        else:
            i = self._cur
        x = self._x[1][i] if self._explores(i) else self._x[2][i]
        return seq[int(mulhi32(x, n))]
