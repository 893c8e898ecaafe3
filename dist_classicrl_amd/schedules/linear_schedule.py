"""Import-path compatibility with the reference (``dist_classicrl.schedules.linear_schedule``)."""

from . import LinearSchedule

__all__ = ["LinearSchedule"]
