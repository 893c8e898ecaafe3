"""Multi-GPU replica synchronisation (replaces the reference's MPI worker tier)."""

from .delta_sync import DeltaSync

__all__ = ["DeltaSync"]
