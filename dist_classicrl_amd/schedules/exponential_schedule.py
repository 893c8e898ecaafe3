"""Import-path compatibility with the reference (``dist_classicrl.schedules.exponential_schedule``)."""

from . import ExponentialSchedule

__all__ = ["ExponentialSchedule"]
