"""Device-resident batched environments (the reference's multi-agent env contract,
``environments/custom_env.py:31-84``, realised in HIP)."""

from .device_envs import DeviceVecEnv, GridLakeEnv, HashTabularEnv, RiggedTwoArmedBanditVecEnv, TicTacToeEnv

__all__ = ["DeviceVecEnv", "GridLakeEnv", "HashTabularEnv", "RiggedTwoArmedBanditVecEnv", "TicTacToeEnv"]
