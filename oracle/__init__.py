"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's hot path.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
`oracle.lib()` loads oracle/_build/libmz_oracle.so, building it with gcc when absent.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "libmz_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("tree_oracle.c", "c4_oracle.c", "data_oracle.c")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB
