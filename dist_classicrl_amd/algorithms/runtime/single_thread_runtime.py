"""Import-path compatibility with the reference (``...runtime.single_thread_runtime``)."""

from .gpu_rollout_runtime import GpuRolloutQLearning, SingleThreadQLearning

__all__ = ["GpuRolloutQLearning", "SingleThreadQLearning"]
