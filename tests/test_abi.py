"""CPU: the C-ABI library loads and exports every symbol ``include/qlearn_engine.h`` declares, the
ctypes table covers exactly those symbols, and the product path refuses to run without a GPU."""

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "qlearn_engine.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qe_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from dist_classicrl_amd import _lib

    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert sorted(_lib.PROTOTYPES) == names
    assert _lib.load().qe_abi_version() == _lib.ABI_VERSION == 2


def test_struct_layouts_match_the_header():
    from dist_classicrl_amd import _lib

    assert ctypes.sizeof(_lib.EnvParams) == 32
    assert ctypes.sizeof(_lib.RolloutStats) == 104


def test_no_cpu_fallback():
    """Without a HIP device the engine must fail loudly (never fall back to the oracle)."""
    import subprocess
    import sys

    code = (
        "import ctypes, sys\n"
        "from dist_classicrl_amd import _lib\n"
        "lib = _lib.load(); h = ctypes.c_void_p()\n"
        "rc = lib.qe_create(ctypes.byref(h), 10, 4, 0.9, 0, 0, 0)\n"
        "print(rc, lib.qe_last_error().decode())\n"
    )
    env = {"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "", "PATH": "/usr/bin:/bin"}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rc = int(out.stdout.split()[0])
    assert rc == -2 and "no HIP device" in out.stdout


def test_product_never_imports_the_oracle():
    for path in (ROOT / "dist_classicrl_amd").rglob("*.py"):
        src = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
    for path in (ROOT / "dist_classicrl_amd" / "csrc").glob("*"):
        if path.suffix in (".h", ".hip", ".cpp"):
            assert "oracle/" not in path.read_text().replace("oracle/draws.py", "").replace("oracle/envs.py", ""), path
