"""Host logic: the schedule classes mirror the reference's (`schedules/*.py`) and `advance_values`
-- the vectorised form the fused rollout uses -- yields exactly the sequence `get_value(); update(n)`
would, including the state it leaves behind."""

import numpy as np
import pytest

from dist_classicrl_amd.schedules import BaseSchedule, ConstantSchedule, ExponentialSchedule, LinearSchedule


def test_reference_update_rules():
    e = ExponentialSchedule(1.0, 0.01, 0.5)  # exponential_schedule.py:22-31
    e.update(2)
    assert e.get_value() == 0.25
    e.update(10)
    assert e.get_value() == 0.01
    lin = LinearSchedule(1.0, -0.1)  # linear_schedule.py:22-31
    lin.update(3)
    assert lin.get_value() == pytest.approx(0.7) and lin.get_value() == 1.0 + 3 * -0.1
    c = ConstantSchedule(0.3)  # constant_schedule.py:6-13
    c.update(1000)
    assert c.get_value() == 0.3


@pytest.mark.parametrize("seed", range(4))
def test_advance_values_equals_the_sequential_loop_bit_for_bit(seed):
    rng = np.random.default_rng(seed)
    for _ in range(400):
        v0 = float(rng.choice([1.0, 0.1, 0.5, rng.random(), 1e-3, 0.0]))
        lo = float(rng.choice([0.01, 1e-5, 0.0, v0, rng.random() * 0.5, 2.0]))
        d = float(rng.choice([0.995, 0.9999, 1.0, 0.5, rng.random(), 0.0, 1.001]))
        n, count = int(rng.choice([1, 2, 128, 4096, 65536])), int(rng.integers(0, 300))
        for make in (lambda: ExponentialSchedule(v0, lo, d), lambda: LinearSchedule(v0, -d * 1e-6),
                     lambda: ConstantSchedule(v0)):
            fast, slow = make(), make()
            got = fast.advance_values(n, count)
            want = BaseSchedule.advance_values(slow, n, count)  # get_value(); update(n); ...
            assert got.dtype == np.float64 and np.array_equal(got, want)
            assert fast.get_value() == slow.get_value()
