"""Mixed-radix codes between multi-discrete vectors and flat state ids -- the host-side counterpart of the
encoding the device TicTacToe environment applies to its board (reference: ``utils.py:12-118``, used by
``wrappers/flatten_multidiscrete_wrapper.py:106-161``).  Most significant digit first: for ``nvec =
[3] * 9`` the radix is ``[6561, 2187, 729, 243, 81, 27, 9, 3, 1]`` and the empty board is state 0."""

from __future__ import annotations

import numpy as np


def compute_radix(nvec):
    """Place values of the digits of ``nvec`` (``utils.py:12-29``)."""
    nvec = np.asarray(nvec)
    place = np.ones_like(nvec)
    place[:-1] = np.cumprod(nvec[:0:-1], dtype=np.int32)[::-1]
    return place.astype(np.int32)


def encode_multi_discrete(multidiscrete_vector, radix) -> int:
    """One vector -> its state id (``utils.py:32-49``)."""
    return int(np.dot(multidiscrete_vector, radix))


def encode_multi_discretes(multidiscrete_vectors, radixes):
    """Rows of vectors -> state ids (``utils.py:52-69``)."""
    return np.sum(np.asarray(multidiscrete_vectors) * radixes, axis=1)


def decode_to_multi_discrete(nvec, index: int, radix):
    """State id -> vector (``utils.py:72-91``)."""
    return (index // np.asarray(radix)) % np.asarray(nvec)


def decode_to_multi_discretes(nvecs, indices, radixes):
    """State ids (column vector or broadcastable) -> vectors (``utils.py:94-113``)."""
    return (np.asarray(indices) // np.asarray(radixes)) % np.asarray(nvecs)
