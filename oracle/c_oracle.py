"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of ``oracle/qlearn_oracle.c`` (closed loop on the
HashTabularEnv).  Used by tests as the fast full-size checker and by ``bench.py``'s ``cpu_baseline``
leg for the "compiled C, one core" figure.  The product never imports this module."""

from __future__ import annotations

import ctypes as C
import subprocess
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libqlearn_oracle.so"


class Cfg(C.Structure):
    _fields_ = [
        ("S", C.c_int64), ("A", C.c_int32), ("n", C.c_int32), ("masked", C.c_int32),
        ("env_seed", C.c_uint32), ("p_term_256", C.c_int32), ("agent_offset", C.c_uint32),
        ("seed", C.c_uint64), ("gamma", C.c_double), ("dtype", C.c_int32), ("mode", C.c_int32),
    ]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB.exists():
            subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
        _lib = C.CDLL(str(LIB))
        _lib.oc_rollout.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CHashRollout:
    """State of one closed-loop run (table, env, bookkeeping) on the C oracle."""

    def __init__(self, n, S, A, *, masked=False, env_seed=1, p_term_256=13, agent_offset=0, seed=0,
                 gamma=0.99, dtype=np.float32, mode="iter"):
        self.lib = load()
        self.cfg = Cfg(S, A, n, int(masked), env_seed, p_term_256, agent_offset, seed, gamma,
                       0 if np.dtype(dtype) == np.float32 else 1, 0 if mode == "iter" else 1)
        self.q = np.zeros((S, A), dtype=dtype)
        self.obs = np.zeros(n, dtype=np.int32)
        self.episode = np.zeros(n, dtype=np.uint32)
        self.acc = np.zeros(n, dtype=np.float32)
        self.step = 0
        self.lib.oc_reset(C.byref(self.cfg), _p(self.obs), _p(self.episode), _p(self.acc))

    def run(self, eps, lr, *, trace=False, log_episodes=True, delta_log=False):
        """``delta_log=True`` adds ``"cells"`` / ``"deltas"``: one (cell = s * A + a, float32 increment) record
        per agent and step, in (step, agent) order -- the content of the engine's delta log."""
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        lr = np.ascontiguousarray(lr, dtype=np.float64)
        steps, n = eps.size, self.cfg.n
        if delta_log:
            d_cells, d_deltas = np.empty(steps * n, dtype=np.uint32), np.empty(steps * n, dtype=np.float32)
            self.lib.oc_set_delta_log(_p(d_cells), _p(d_deltas), C.c_int64(steps * n))
        tr = np.empty((steps, n), dtype=np.int32) if trace else None
        cap = steps * n if log_episodes else 0
        es, ea = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32)
        er = np.empty(cap, dtype=np.float32)
        cnt = C.c_int64(0)
        rc = self.lib.oc_rollout(C.byref(self.cfg), _p(self.q), _p(self.obs), _p(self.episode), _p(self.acc),
                                 C.c_uint64(self.step), C.c_int64(steps), _p(eps), _p(lr), _p(tr), _p(es), _p(ea),
                                 _p(er), C.c_int64(cap), C.byref(cnt))
        if rc:  # like the reference: random.choice on an empty candidate set (q_learning_optimal.py:470, :563)
            self.step += rc - 1
            self.failed_step = rc - 1
            msg = "Cannot choose from an empty sequence"
            raise IndexError(msg)
        self.step += steps
        k = min(cnt.value, cap)
        out = {"actions": tr, "history": er[:k].copy(), "ep_step": es[:k].copy(), "episodes": cnt.value}
        if delta_log:
            self.lib.oc_set_delta_log(None, None, C.c_int64(0))
            out["cells"], out["deltas"] = d_cells, d_deltas
        return out


def exp_schedule(v0, vmin, decay, n_updates, count):
    """Values an ExponentialSchedule yields at `count` consecutive vector steps (schedules/exponential_schedule.py:31)."""
    out, v, f = np.empty(count), v0, decay**n_updates
    for t in range(count):
        out[t] = v
        v = max(v * f, vmin)
    return out, v


def time_rollout(wl, seconds=3.0):
    """env-steps/s of the compiled loop on one core, same workload/schedules as bench.py."""
    n = wl["agents"]
    run = CHashRollout(n, wl["states"], wl["actions"], masked=wl["masked"], dtype=np.float64)
    run.q.fill(0.0)
    block = max(10, 200_000 // n)
    eps, e_last = exp_schedule(1.0, 0.01, 0.995, n, block)
    lr, l_last = exp_schedule(0.1, 1e-5, 0.995, n, block)
    run.run(eps, lr, log_episodes=False)  # warm-up; schedules are at their floors afterwards
    eps[:], lr[:] = e_last, l_last
    rates, t_start = [], time.perf_counter()
    while time.perf_counter() - t_start < seconds or not rates:
        t0 = time.perf_counter()
        run.run(eps, lr, log_episodes=False)
        rates.append(block * n / (time.perf_counter() - t0))
    return float(np.median(rates))
