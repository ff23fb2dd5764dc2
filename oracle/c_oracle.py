"""ORACLE (test infrastructure): ctypes binding of oracle/_build/liboracle.so
(the plain-C kNN restatement, oracle/knn_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "knn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.gfs3d_oracle_knn_f32.restype = ctypes.c_int
        _lib.gfs3d_oracle_knn_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def knn(x: np.ndarray, k: int, return_dist: bool = False):
    """x: (B,C,N) float32 -> idx (B,N,k) int64 [, pd (B,N,k) float32]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, C, N = x.shape
    idx = np.empty((B, N, k), dtype=np.int64)
    dist = np.empty((B, N, k), dtype=np.float32) if return_dist else None
    rc = _load().gfs3d_oracle_knn_f32(x.ctypes.data, B, C, N, k, idx.ctypes.data,
                                      dist.ctypes.data if return_dist else None)
    if rc != 0:
        raise RuntimeError(f"gfs3d_oracle_knn_f32 failed rc={rc}")
    return (idx, dist) if return_dist else idx
