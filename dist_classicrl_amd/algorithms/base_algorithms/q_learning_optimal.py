"""``OptimalQLearningBase`` with the Q-table resident in MI355X HBM.

Host-side mirror of the reference class
(``dist_classicrl/algorithms/base_algorithms/q_learning_optimal.py:23-934``): same constructor,
attributes and method names, so it can be used wherever the reference class is.  All arithmetic runs
in ``libqlearn_engine.so`` (``include/qlearn_engine.h``); this file only marshals arrays.

Differences a caller can observe (all documented in DESIGN.md):

* the table dtype is chosen at construction (``dtype=np.float32`` by default, ``np.float64`` gives
  the reference's default precision); assigning ``q_table`` casts into that dtype.
* ``q_table`` returns a *snapshot* that writes through to the device on item assignment
  (``algo.q_table[s] = row`` works as in the reference's tests); it does not track later updates.
* randomness is counter based (Philox4x32-10 keyed by ``seed``; one step index per
  ``choose_actions`` call -- see ``oracle/draws.py`` for the protocol) instead of the reference's
  two sequential generators; every reference selection variant shares one distribution, so all the
  variant method names map onto one kernel family.
"""

from __future__ import annotations

import ctypes as C
import secrets

import numpy as np

from dist_classicrl_amd import _lib

# dispatcher thresholds of the reference (q_learning_optimal.py:14-20); only used to reproduce
# which calls return -1 and which raise IndexError when an agent has no selectable action.
DETERMINISTIC_MAX_ACTION_SIZE_ITER = 10
DETERMINISTIC_MIN_ACTION_SIZE_VEC_ITER = 10000
DETERMINISTIC_MAX_NUM_STATES_VEC_ITER = 3
NO_ACTION_MASKS_NO_DETERMINISTIC_MAX_NUM_STATES_ITER = 100
ACTION_MASKS_NO_DETERMINISTIC_MAX_ACTION_SIZE_ITER = 10


class _QTableSnapshot(np.ndarray):
    """(S, A) copy of the device table whose item assignments are written through to HBM."""

    _owner = None

    def __array_finalize__(self, obj):
        self._owner = None  # views/derived arrays are plain host data

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        if self._owner is not None:
            self._owner._write_through(self, key)


class OptimalQLearningBase:
    """Tabular Q-learning core on one MI355X (reference :23-98 for the constructor contract)."""

    def __init__(self, state_size, action_size, discount_factor=0.97, seed=None, *,
                 dtype=np.float32, device=0):
        self.state_size = int(state_size)
        self.action_size = int(action_size)
        self.discount_factor = discount_factor
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            msg = "dtype must be float32 or float64"
            raise ValueError(msg)
        self.seed = secrets.randbits(64) if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self.device = int(device)
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.qe_create(
            C.byref(self._h), self.state_size, self.action_size, float(discount_factor), self.seed,
            _lib.QE_F32 if self.dtype == np.float32 else _lib.QE_F64, self.device))
        self._gamma_sent = float(discount_factor)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.qe_destroy(h)
            self._h = C.c_void_p()

    # ------------------------------------------------------------------ plumbing
    @property
    def handle(self):
        """The ``qe_engine*`` (for runtimes / environments living on the same GPU)."""
        return self._h

    @property
    def step_counter(self) -> int:
        """Index of the next vector step in the draw protocol."""
        return int(self._lib.qe_get_step_counter(self._h))

    @step_counter.setter
    def step_counter(self, value: int) -> None:
        _lib.check(self._lib.qe_set_step_counter(self._h, int(value)))

    def set_rollout_path(self, path: str) -> None:
        """Tuning knob, never changes results: ``"auto"``, ``"stepwise"`` (one kernel pair per vector
        step), ``"persistent"`` (one launch per rollout; needs <= 512 agents and agents x lanes-per-row
        <= 1024), ``"wide"`` (step-wise, the ordered path spread over the whole chip) or ``"turnstile"``
        (one launch per vector step, shared rows handed from agent to agent inside it: ``learn_iter``
        with up to ~60 000 agents; elsewhere the automatic choice applies)."""
        code = {"auto": _lib.PATH_AUTO, "stepwise": _lib.PATH_STEPWISE, "persistent": _lib.PATH_PERSISTENT,
                "wide": _lib.PATH_WIDE, "turnstile": _lib.PATH_TURNSTILE}[path]
        _lib.check(self._lib.qe_set_option(self._h, _lib.OPT_ROLLOUT_PATH, code))

    def set_engine_option(self, option: int, value: int) -> None:
        """Other tuning knobs of ``include/qlearn_engine.h`` (``_lib.OPT_*``); none changes results."""
        _lib.check(self._lib.qe_set_option(self._h, int(option), int(value)))

    @property
    def lanes_per_row(self) -> int:
        """Lanes of a wavefront that share one Q-table row (power of two, 4 columns per lane)."""
        ld, lanes = int(self._lib.qe_table_row_stride(self._h)), 1
        while 4 * lanes < ld and lanes < 64:
            lanes *= 2
        return lanes

    def _qe_dtype(self, dt):
        return _lib.QE_F32 if np.dtype(dt) == np.float32 else _lib.QE_F64

    # ------------------------------------------------------------------ table access (:96-261)
    @property
    def q_table(self):
        host = np.empty((self.state_size, self.action_size), dtype=self.dtype)
        _lib.check(self._lib.qe_table_download(self._h, host.ctypes.data, self._qe_dtype(self.dtype)))
        snap = host.view(_QTableSnapshot)
        snap._owner = self
        return snap

    @q_table.setter
    def q_table(self, value):
        arr = np.asarray(value)
        if arr.shape != (self.state_size, self.action_size):
            msg = f"q_table must have shape {(self.state_size, self.action_size)}, got {arr.shape}"
            raise ValueError(msg)
        up_dt = np.float32 if arr.dtype == np.float32 else np.float64
        arr = np.ascontiguousarray(arr, dtype=up_dt)
        _lib.check(self._lib.qe_table_upload(self._h, arr.ctypes.data, self._qe_dtype(up_dt)))

    def _write_through(self, snap, key):
        first = key[0] if isinstance(key, tuple) else key
        if isinstance(first, (int, np.integer)) and self.state_size * self.action_size > (1 << 16):
            s = int(first) % self.state_size  # one row changed: send only that row
            self._cells(np.full(self.action_size, s), np.arange(self.action_size), np.asarray(snap)[s], 1)
        else:
            self.q_table = np.asarray(snap)

    def _cells(self, states, actions, values, op):
        states, actions = _lib.as_i32(states).ravel(), _lib.as_i32(actions).ravel()
        if states.shape != actions.shape:
            msg = "states and actions must have the same length"
            raise ValueError(msg)
        vals = np.empty(states.size, dtype=np.float64) if op == 0 else np.ascontiguousarray(
            np.broadcast_to(np.asarray(values, dtype=np.float64).ravel(), states.shape))
        _lib.check(self._lib.qe_table_cells(
            self._h, _lib.ptr(states, C.c_int32), _lib.ptr(actions, C.c_int32), states.size,
            _lib.ptr(vals, C.c_double), op))
        return vals

    def get_q_value(self, state, action):
        return self.dtype.type(self._cells([state], [action], None, 0)[0])

    def get_q_values(self, states, actions):
        return self._cells(states, actions, None, 0).astype(self.dtype)

    def get_state_q_values(self, state):
        a = np.arange(self.action_size)
        return self._cells(np.full(self.action_size, state), a, None, 0).astype(self.dtype)

    def get_states_q_values(self, states):
        states = np.asarray(states).ravel()
        s = np.repeat(states, self.action_size)
        a = np.tile(np.arange(self.action_size), states.size)
        return self._cells(s, a, None, 0).astype(self.dtype).reshape(states.size, self.action_size)

    def get_action_q_values(self, action):
        return np.asarray(self.q_table)[:, action]

    def get_actions_q_values(self, actions):
        return np.asarray(self.q_table)[:, np.asarray(actions)]

    def set_q_value(self, state, action, value):
        self._cells([state], [action], [value], 1)

    def add_q_value(self, state, action, value):
        self._cells([state], [action], [value], 2)

    def add_q_values(self, states, actions, values):
        """``np.add.at`` semantics: duplicates accumulate, in index order (reference :235-250)."""
        self._cells(states, actions, values, 2)

    _IO_CHUNK_BYTES = 64 << 20  # rows are streamed between HBM and the file in blocks of this size

    def save(self, filename):
        """The (S, A) table as a ``.npy`` file, the reference's on-disk format (``np.save``, :252-261),
        streamed from HBM block by block: no second copy of the table on the host."""
        filename = str(filename)
        if not filename.endswith(".npy"):
            filename += ".npy"  # np.save appends the suffix too
        rows_per = max(1, self._IO_CHUNK_BYTES // (self.action_size * self.dtype.itemsize))
        with open(filename, "wb") as f:
            np.lib.format.write_array_header_1_0(f, {"descr": np.lib.format.dtype_to_descr(self.dtype),
                                                      "fortran_order": False,
                                                      "shape": (self.state_size, self.action_size)})
            block = np.empty((min(rows_per, self.state_size), self.action_size), dtype=self.dtype)
            for first in range(0, self.state_size, rows_per):
                k = min(rows_per, self.state_size - first)
                _lib.check(self._lib.qe_table_download_rows(self._h, block.ctypes.data, first, k))
                f.write(memoryview(block[:k]))

    def load(self, filename):
        """Counterpart of :meth:`save` (the reference has none: its users assign ``np.load(...)`` to
        ``q_table``): streams a ``.npy`` table of this shape into HBM, casting to the table's dtype."""
        src = np.load(filename, mmap_mode="r")
        if src.shape != (self.state_size, self.action_size):
            msg = f"{filename} holds a table of shape {src.shape}, expected {(self.state_size, self.action_size)}"
            raise ValueError(msg)
        rows_per = max(1, self._IO_CHUNK_BYTES // (self.action_size * self.dtype.itemsize))
        for first in range(0, self.state_size, rows_per):
            block = np.ascontiguousarray(src[first:first + rows_per], dtype=self.dtype)
            _lib.check(self._lib.qe_table_upload_rows(self._h, block.ctypes.data, first, block.shape[0]))

    # ------------------------------------------------------------------ selection (:263-726)
    def _select(self, states, exploration_rate, deterministic, action_masks, numpy_variant=False, numpy_max=None):
        # numpy_variant: the NumPy variants' treatment of an all-zero mask; numpy_max: np.max as the row maximum
        # (NaN-propagating; every NumPy variant) instead of the list variants' scan (steps over NaN)
        numpy_max = numpy_variant if numpy_max is None else numpy_max
        states = _lib.as_i32(states).ravel()
        n = states.size
        masks = None
        if action_masks is not None:
            masks = _lib.as_u8_flags(action_masks)
            assert masks.shape == (n, self.action_size), (
                "Action masks must match the number of states and actions."
            )
        out = np.empty(n, dtype=np.int32)
        _lib.check(self._lib.qe_choose_actions(
            self._h, _lib.ptr(states, C.c_int32), n, _lib.ptr(masks, C.c_uint8),
            float(exploration_rate), (1 if deterministic else 0) | (2 if numpy_variant else 0) | (4 if numpy_max else 0),
            _lib.ptr(out, C.c_int32)))
        return out

    def _list_variant(self, n, deterministic, masked):
        """True when the reference dispatcher (:644-726) would run a Python-list variant, which
        returns -1 for an agent without selectable action; the NumPy variants raise IndexError."""
        if deterministic:
            return self.action_size <= DETERMINISTIC_MAX_ACTION_SIZE_ITER
        if not masked:
            return n < NO_ACTION_MASKS_NO_DETERMINISTIC_MAX_NUM_STATES_ITER
        return self.action_size <= ACTION_MASKS_NO_DETERMINISTIC_MAX_ACTION_SIZE_ITER

    def choose_actions(self, states, exploration_rate, *, deterministic=False, action_masks=None):
        numpy_variant = not self._list_variant(np.size(states), deterministic, action_masks is not None)
        out = self._select(states, exploration_rate, deterministic, action_masks, numpy_variant)
        if numpy_variant and (out < 0).any():
            msg = "Cannot choose from an empty sequence"
            raise IndexError(msg)
        return out

    def choose_actions_iter(self, states, exploration_rate, *, deterministic=False, action_masks=None):
        return self._select(states, exploration_rate, deterministic, action_masks)

    def choose_actions_vec_iter(self, states, exploration_rate, *, deterministic=False, action_masks=None):
        return self._raise_on_empty(self._select(states, exploration_rate, deterministic, action_masks, True))

    def choose_actions_vec(self, states, exploration_rate, *, deterministic=False):
        return self._raise_on_empty(self._select(states, exploration_rate, deterministic, None, numpy_max=True))

    def choose_masked_actions_vec(self, states, action_masks, exploration_rate, *, deterministic=False):
        return self._raise_on_empty(self._select(states, exploration_rate, deterministic, action_masks, True))

    @staticmethod
    def _raise_on_empty(out):
        if (out < 0).any():
            msg = "Cannot choose from an empty sequence"
            raise IndexError(msg)
        return out

    def choose_action(self, state, exploration_rate, *, deterministic=False):
        return int(self._select([state], exploration_rate, deterministic, None)[0])

    def choose_masked_action(self, state, action_mask, exploration_rate, *, deterministic=False):
        assert len(action_mask) == self.action_size, (
            "Action mask should have the same length as the action size."
        )
        return int(self._select([state], exploration_rate, deterministic, np.asarray(action_mask)[None, :])[0])

    def choose_action_vec(self, state, exploration_rate, *, deterministic=False):
        return int(self._raise_on_empty(self._select([state], exploration_rate, deterministic, None, numpy_max=True))[0])

    def choose_masked_action_vec(self, state, action_mask, exploration_rate, *, deterministic=False):
        mask = np.asarray(list(action_mask))
        assert mask.size == self.action_size, "Action mask should have the same size as the action space."
        return int(self._raise_on_empty(self._select([state], exploration_rate, deterministic, mask[None, :], True))[0])

    # ------------------------------------------------------------------ learning (:728-934)
    def _learn(self, states, actions, rewards, next_states, terminated, lr, next_action_masks, mode):
        if float(self.discount_factor) != self._gamma_sent:
            msg = "discount_factor is fixed at construction on the device engine"
            raise ValueError(msg)
        states, actions = _lib.as_i32(states).ravel(), _lib.as_i32(actions).ravel()
        next_states = _lib.as_i32(next_states).ravel()
        rewards = np.ascontiguousarray(rewards, dtype=np.float32).ravel()
        terminated = _lib.as_u8_flags(terminated).ravel()
        n = states.size
        if not (actions.size == rewards.size == next_states.size == terminated.size == n):
            msg = "zip() arguments have different lengths"  # reference: zip(strict=True), :802
            raise ValueError(msg)
        masks = None
        if next_action_masks is not None:
            masks = _lib.as_u8_flags(next_action_masks)
            if masks.shape != (n, self.action_size):
                msg = "next_action_masks must have shape (n, action_size)"
                raise ValueError(msg)
        _lib.check(self._lib.qe_learn(
            self._h, _lib.ptr(states, C.c_int32), _lib.ptr(actions, C.c_int32),
            _lib.ptr(rewards, C.c_float), _lib.ptr(next_states, C.c_int32),
            _lib.ptr(terminated, C.c_uint8), n, float(lr), _lib.ptr(masks, C.c_uint8), mode))

    def learn(self, states, actions, rewards, next_states, terminated, lr, next_action_masks=None):
        """Sequential semantics, like the reference dispatcher (:893-934)."""
        self._learn(states, actions, rewards, next_states, terminated, lr, next_action_masks, _lib.LEARN_ITER)

    def learn_iter(self, states, actions, rewards, next_states, terminated, lr, next_action_masks=None):
        self._learn(states, actions, rewards, next_states, terminated, lr, next_action_masks, _lib.LEARN_ITER)

    def learn_vec(self, states, actions, rewards, next_states, terminated, lr, next_action_masks=None):
        """Batch semantics: all reads precede all writes; colliding updates accumulate (:819-891)."""
        self._learn(states, actions, rewards, next_states, terminated, lr, next_action_masks, _lib.LEARN_VEC)

    def _learn_vec(self, states, actions, rewards, next_states, terminated, lr, next_action_masks=None):
        """The reference's private worker behind ``learn_vec`` (:853-891); same thing here."""
        self.learn_vec(states, actions, rewards, next_states, terminated, lr, next_action_masks)

    def single_learn(self, state, action, reward, next_state, terminated, lr, next_action_mask=None):
        masks = None if next_action_mask is None else np.asarray(next_action_mask)[None, :]
        self._learn([state], [action], [reward], [next_state], [terminated], lr, masks, _lib.LEARN_ITER)


HipQLearning = OptimalQLearningBase
