"""Exploration-rate / learning-rate schedules with the reference's interface
(``schedules/base_schedules.py:27-74``): ``get_value()``, ``set_value(v)``, ``update(steps)``.

``update(steps)`` is called once per vector step with ``steps = number of agents``
(``algorithms/runtime/base_runtime.py:262-263``).  The fused GPU rollout needs the values of many
consecutive vector steps up front; :meth:`BaseSchedule.advance_values` produces exactly the
sequence ``get_value(); update(n); get_value(); ...`` would and leaves the schedule advanced.
All arithmetic is Python ``float`` (double), like the reference.
"""

from __future__ import annotations

import numpy as np


class BaseSchedule:
    def __init__(self, value: float, min_value: float) -> None:
        self.value = value
        self.min_value = min_value

    def get_value(self) -> float:
        return self.value

    def set_value(self, value: float) -> None:
        self.value = value

    def update(self, steps: int) -> None:
        msg = "update() is implemented by the concrete schedules"
        raise NotImplementedError(msg)

    def advance_values(self, n_updates: int, count: int) -> np.ndarray:
        """Values read at ``count`` consecutive vector steps (float64); advances the schedule."""
        out = np.empty(count, dtype=np.float64)
        for t in range(count):
            out[t] = self.get_value()
            self.update(n_updates)
        return out

    def advance_into(self, n_updates: int, count: int, out) -> None:
        """The same values written into ``out[0:count]`` (any indexable buffer, e.g. a ctypes array): the
        form a short fused rollout uses, no NumPy set-up cost."""
        for t in range(count):
            out[t] = self.get_value()
            self.update(n_updates)


class ConstantSchedule(BaseSchedule):
    """``schedules/constant_schedule.py:6-13``."""

    def __init__(self, value: float) -> None:
        super().__init__(value, value)

    def update(self, steps: int) -> None:
        return None

    def advance_values(self, n_updates: int, count: int) -> np.ndarray:
        return np.full(count, self.get_value(), dtype=np.float64)

    def advance_into(self, n_updates: int, count: int, out) -> None:
        out[0:count] = [self.get_value()] * count


class ExponentialSchedule(BaseSchedule):
    """``v <- max(v * decay_rate**steps, min_value)`` (``schedules/exponential_schedule.py:22-31``)."""

    def __init__(self, value: float, min_value: float, decay_rate: float) -> None:
        super().__init__(value, min_value)
        self.decay_rate = decay_rate

    def update(self, steps: int) -> None:
        self.set_value(max(self.get_value() * (self.decay_rate**steps), self.min_value))

    def advance_into(self, n_updates: int, count: int, out) -> None:
        # the same float64 operations in the same order as `update`, without a method call per step
        v, lo, f = self.get_value(), self.min_value, self.decay_rate**n_updates
        if v == lo and 0.0 <= f <= 1.0 and lo >= 0.0:  # at the floor: max(lo * f, lo) == lo from here on
            out[0:count] = [lo] * count
            return
        for t in range(count):
            out[t] = v
            v = max(v * f, lo)
        self.set_value(v)

    def advance_values(self, n_updates: int, count: int) -> np.ndarray:
        v, lo, f = float(self.get_value()), float(self.min_value), float(self.decay_rate**n_updates)
        if count > 0 and 0.0 <= f <= 1.0 and lo >= 0.0:
            # The recurrence is a running product until it first falls to the floor and the floor from
            # then on (lo * f <= lo).  multiply.accumulate performs the same float64 products in the
            # same order as the loop below, so the values are identical bit for bit.
            prod = np.multiply.accumulate(np.concatenate(([v], np.full(count, f, dtype=np.float64))))
            below = np.flatnonzero(prod[1:] <= lo)
            k = int(below[0]) + 1 if below.size else count + 1  # first step that reads the floor
            out = prod[:count].copy()
            out[k:] = lo
            self.set_value(lo if k <= count else float(prod[count]))
            return out
        out = np.empty(count, dtype=np.float64)
        for t in range(count):  # general case (growing or negative schedules), sequential
            out[t] = v
            v = max(v * f, lo)
        self.set_value(v)
        return out


class LinearSchedule(BaseSchedule):
    """``v <- v + steps * decay_rate``, unclamped (``schedules/linear_schedule.py:22-31``)."""

    def __init__(self, value: float, decay_rate: float) -> None:
        super().__init__(value, -1e9)
        self.decay_rate = decay_rate

    def update(self, steps: int) -> None:
        self.set_value(self.get_value() + steps * self.decay_rate)

    def advance_into(self, n_updates: int, count: int, out) -> None:
        v, inc = self.get_value(), n_updates * self.decay_rate
        for t in range(count):
            out[t] = v
            v = v + inc
        self.set_value(v)

    def advance_values(self, n_updates: int, count: int) -> np.ndarray:
        v, inc = float(self.get_value()), float(n_updates * self.decay_rate)
        # add.accumulate performs the same float64 additions in the same order as `update`
        run = np.add.accumulate(np.concatenate(([v], np.full(count, inc, dtype=np.float64))))
        self.set_value(float(run[count]))
        return run[:count].copy()


__all__ = ["BaseSchedule", "ConstantSchedule", "ExponentialSchedule", "LinearSchedule"]
