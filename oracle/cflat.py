"""ctypes loader for the plain-C oracle (oracle/flat_search.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_flat.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "flat_search.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        # -march=native is unsafe for a .so that travels to another box: build generic x86-64-v3
        subprocess.check_call(["make", "-C", _HERE, "-s",
                               "CFLAGS=-O3 -mavx2 -mfma -fopenmp -fPIC -std=c11 -Wall -Wextra"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_flat_search.restype = ctypes.c_int
        _lib.oracle_flat_search.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def flat_search(q: np.ndarray, x: np.ndarray, k: int, metric: int = 0, alive: np.ndarray | None = None):
    q = np.ascontiguousarray(q, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    nq, d = q.shape
    out_s = np.empty((nq, k), dtype=np.float32)
    out_r = np.empty((nq, k), dtype=np.int64)
    am = None
    if alive is not None:
        am = np.ascontiguousarray(alive, dtype=np.uint8)
    rc = lib().oracle_flat_search(q.ctypes.data, nq, x.ctypes.data, x.shape[0], d, k, metric,
                                  am.ctypes.data if am is not None else None,
                                  out_s.ctypes.data, out_r.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_flat_search failed: {rc}")
    return out_s, out_r
