"""ctypes binding of ``libqlearn_engine.so`` (C ABI: ``include/qlearn_engine.h``).

There is deliberately no CPU fallback: if the HIP library is missing, or no MI355X is visible,
every compute entry point raises.  Building the library: ``python __graft_entry__.py`` or
``dist_classicrl_amd/csrc/build.sh``.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import os

# QE_LIB_PATH lets a developer A/B two builds of the library in one session; default = in-tree build
LIB_PATH = Path(os.environ.get("QE_LIB_PATH") or Path(__file__).resolve().parent / "csrc" / "libqlearn_engine.so")

ABI_VERSION = 2  # QE_ABI_VERSION of include/qlearn_engine.h
QE_F32, QE_F64 = 0, 1
LEARN_ITER, LEARN_VEC = 0, 1
ENV_HASH, ENV_GRID, ENV_BANDIT, ENV_TICTACTOE = 0, 1, 2, 3
OPT_ROLLOUT_PATH = 0
OPT_USE_GRAPH = 1
OPT_TOKEN_ROUNDS = 2
OPT_LISTED_MIN_AGENTS = 3
OPT_EVENT_TIMING = 4
OPT_HOST_BLOCK = 5
OPT_LANE_ORDERED_PATH = 6
OPT_TURN_FORWARD = 7
OPT_TURN_POLL = 8
OPT_STAMP_HASH_BITS = 9
PATH_AUTO, PATH_STEPWISE, PATH_PERSISTENT, PATH_WIDE, PATH_TURNSTILE = 0, 1, 2, 3, 4

ERR_INVALID, ERR_NO_DEVICE, ERR_OOM, ERR_UNSUPPORTED, ERR_INDEX = -1, -2, -3, -4, -5


def decode_variant(v: int) -> dict:
    """Fields of ``qe_rollout_stats.kernel_variant`` (include/qlearn_engine.h)."""
    v = int(v)
    return {
        "path": {1: "stepwise", 2: "persistent", 3: "wide", 4: "turnstile", 5: "eval"}.get(v & 15, "none"),
        "lean": (v >> 4) & 3, "help": bool((v >> 6) & 1), "full": bool((v >> 7) & 1), "light": bool((v >> 8) & 1),
        "cap512": bool((v >> 9) & 1), "dataflow": bool((v >> 10) & 1), "nv": (v >> 12) & 255, "masked": bool((v >> 20) & 1),
    }


def variant_symbol(v: int, dtype: str = "float", env: str = "HashEnv", lanes_per_row: int = 4, vec: bool = False) -> str:
    """The kernel instantiation behind ``kernel_variant`` as rocprofv3 prints it (without the ``void qe::`` prefix and
    the argument list): what ``bench.py`` names in ``roofline.kernel`` and matches profile files against."""
    d = decode_variant(v)
    b = lambda x: "true" if x else "false"  # noqa: E731
    if d["path"] == "persistent" and d["dataflow"]:
        return f"k_rollout_df<{dtype}, qe::{env}, {d['nv']}, {b(d['masked'])}, {d['lean']}, {b(d['full'])}>"
    if d["path"] == "persistent":
        cap = 512 if d["cap512"] else 128
        return (f"k_rollout_lane<{dtype}, qe::{env}, {d['nv']}, {cap}, {b(d['masked'])}, {d['lean']}, {b(d['help'])}, "
                f"{b(d['full'])}, {b(d['light'])}>")
    lc = lanes_per_row if env == "HashEnv" and lanes_per_row in (4, 8, 16) else 0
    name = {"turnstile": "k_step_turn", "stepwise": "k_step_fast", "wide": "k_step_fast", "eval": "k_eval"}.get(d["path"], "?")
    if name == "k_step_turn":
        return f"k_step_turn<{dtype}, qe::{env}, {lc}, {b(vec)}>"
    return f"{name}<{dtype}, qe::{env}, {lc}>" if name != "k_eval" else f"k_eval<{dtype}, qe::{env}>"


class EngineError(RuntimeError):
    """The HIP engine reported a failure that has no closer Python equivalent."""


class EnvParams(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("masked", C.c_int32),
        ("seed", C.c_uint32),
        ("p_term_256", C.c_int32),
        ("side", C.c_int32),
        ("episode_len", C.c_int32),
        ("agent_offset", C.c_uint32),
        ("reserved", C.c_int32),
    ]


class RolloutStats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_double),
        ("launches", C.c_int64),
        ("episodes", C.c_int64),
        ("involved", C.c_int64),
        ("episodes_dropped", C.c_int64),
        ("dominant_ms", C.c_double),
        ("dominant_launches", C.c_int64),
        ("dominant_env_steps", C.c_int64),
        ("device_clock_ms", C.c_double),
        ("host_begin_us", C.c_double),
        ("host_end_us", C.c_double),
        ("kernel_variant", C.c_int64),
        ("complex_steps", C.c_int64),
    ]


_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_U32P = C.POINTER(C.c_uint32)
_U8P = C.POINTER(C.c_uint8)
_F32P = C.POINTER(C.c_float)
_F64P = C.POINTER(C.c_double)
_I64P = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must list every symbol include/qlearn_engine.h declares
PROTOTYPES = {
    "qe_abi_version": (C.c_int, []),
    "qe_last_error": (C.c_char_p, []),
    "qe_create": (C.c_int, [C.POINTER(_P), C.c_int64, C.c_int32, C.c_double, C.c_uint64, C.c_int32, C.c_int32]),
    "qe_destroy": (C.c_int, [_P]),
    "qe_synchronize": (C.c_int, [_P]),
    "qe_set_stream": (C.c_int, [_P, _P]),
    "qe_set_option": (C.c_int, [_P, C.c_int32, C.c_int64]),
    "qe_table_upload": (C.c_int, [_P, _P, C.c_int32]),
    "qe_table_download": (C.c_int, [_P, _P, C.c_int32]),
    "qe_table_download_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "qe_table_upload_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "qe_table_cells": (C.c_int, [_P, _I32P, _I32P, C.c_int64, _F64P, C.c_int32]),
    "qe_table_dev": (_P, [_P]),
    "qe_table_row_stride": (C.c_int64, [_P]),
    "qe_set_step_counter": (C.c_int, [_P, C.c_uint64]),
    "qe_get_step_counter": (C.c_uint64, [_P]),
    "qe_set_agent_offset": (C.c_int, [_P, C.c_uint32]),
    "qe_choose_actions": (C.c_int, [_P, _I32P, C.c_int64, _U8P, C.c_double, C.c_int32, _I32P]),
    "qe_learn": (C.c_int, [_P, _I32P, _I32P, _F32P, _I32P, _U8P, C.c_int64, C.c_double, _U8P, C.c_int32]),
    "qe_env_create": (C.c_int, [C.POINTER(_P), _P, C.c_int64, C.POINTER(EnvParams)]),
    "qe_env_destroy": (C.c_int, [_P]),
    "qe_env_reset": (C.c_int, [_P, C.c_int32, C.c_uint32]),
    "qe_env_observe": (C.c_int, [_P, _I32P, _U8P, _F32P]),
    "qe_env_restore": (C.c_int, [_P, _I32P, _U32P, _F32P]),
    "qe_env_aux": (C.c_int, [_P, _U32P]),
    "qe_env_step": (C.c_int, [_P, _I32P, _I32P, _F32P, _U8P, _U8P]),
    "qe_rollout": (C.c_int, [_P, _P, C.c_int64, _F64P, _F64P, C.c_int32, _I32P, C.POINTER(RolloutStats)]),
    "qe_rollout_begin": (C.c_int, [_P, _P, C.c_int64, _F64P, _F64P, C.c_int32, C.c_int32]),
    "qe_schedule_plan": (C.c_int, [_P, _F64P, _F64P, C.c_int64]),
    "qe_rollout_end": (C.c_int, [_P, C.c_int32, C.POINTER(RolloutStats)]),
    "qe_rollout_chunk_limit": (C.c_int64, [_P, _P, C.c_int32]),
    "qe_rollout_fused": (C.c_int64, [_P, _P, C.c_int64, _F64P, _F64P, C.c_int32, C.POINTER(RolloutStats), C.c_int64, _I32P,
                                     _F32P, _F32P, _I32P, _U32P, _F32P]),
    "qe_evaluate": (C.c_int, [_P, _P, C.c_int64, C.POINTER(RolloutStats)]),
    "qe_episode_log": (C.c_int64, [_P, C.c_int64, _I32P, _I32P, _F32P]),
    "qe_delta_log_attach": (C.c_int, [_P, _P, C.c_int64]),
    "qe_delta_log_count": (C.c_int64, [_P]),
    "qe_delta_log_reset": (C.c_int, [_P]),
    "qe_delta_apply_dev": (C.c_int, [_P, _P, C.c_int64]),
    "qe_delta_apply_skip_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64]),
    "qe_delta_apply_sorted_dev": (C.c_int, [_P, _P, C.c_int64]),
    "qe_delta_apply_gathered_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "qe_debug_occupy_cus": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "qe_replay_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64]),
    "qe_replay_destroy": (C.c_int, [_P]),
    "qe_replay_push": (C.c_int, [_P, _I64P, _I64P, _F64P, _I64P, _U8P, C.c_int64]),
    "qe_replay_attach": (C.c_int, [_P, _P]),
    "qe_replay_len": (C.c_int64, [_P]),
    "qe_replay_position": (C.c_int64, [_P]),
    "qe_replay_full": (C.c_int32, [_P]),
    "qe_replay_gather": (C.c_int, [_P, _I64P, C.c_int64, _I64P, _I64P, _F64P, _I64P, _U8P]),
    "qe_replay_learn": (C.c_int, [_P, _P, _I64P, C.c_int64, C.c_double, C.c_int32]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises ``ImportError`` with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        msg = (
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or "
            "dist_classicrl_amd/csrc/build.sh). The engine has no CPU fallback."
        )
        raise ImportError(msg)
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table diverge
        fn.restype = res
        fn.argtypes = args
    if lib.qe_abi_version() != ABI_VERSION:
        msg = f"ABI version mismatch: library {lib.qe_abi_version()}, binding {ABI_VERSION}"
        raise ImportError(msg)
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Translate a negative ``qe_status`` into the exception the reference would raise."""
    if rc >= 0:
        return
    text = load().qe_last_error().decode(errors="replace")
    if rc == ERR_INDEX:
        raise IndexError(text)
    if rc == ERR_INVALID:
        raise ValueError(text)
    if rc == ERR_OOM:
        raise MemoryError(text)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    raise EngineError(text)


def ptr(arr, ctype):
    """Typed pointer to a C-contiguous NumPy array (``None`` -> NULL)."""
    if arr is None:
        return None
    return arr.ctypes.data_as(C.POINTER(ctype))


def as_i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def as_u8_flags(x):
    """Truthiness -> one byte per element (the reference treats any non-zero mask entry as valid)."""
    x = np.asarray(x)
    return np.ascontiguousarray(x != 0, dtype=np.uint8)
