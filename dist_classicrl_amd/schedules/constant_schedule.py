"""Import-path compatibility with the reference (``dist_classicrl.schedules.constant_schedule``)."""

from . import ConstantSchedule

__all__ = ["ConstantSchedule"]
