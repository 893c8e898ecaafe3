"""Import-path compatibility with the reference (``dist_classicrl.schedules.base_schedules``)."""

from . import BaseSchedule

__all__ = ["BaseSchedule"]
