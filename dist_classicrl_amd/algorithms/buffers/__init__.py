"""Experience replay on the device (reference: ``algorithms/buffers/`` -- WIP and unused upstream)."""

from .experience_replay import ExperienceReplay

__all__ = ["ExperienceReplay"]
