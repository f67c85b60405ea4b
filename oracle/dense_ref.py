"""Python face of the C inner-product-search oracle (test infrastructure only).

``oracle/dense_topk_ref.c`` holds the arithmetic and its reference citation; this module
builds it with gcc and calls it through ctypes.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "dense_topk_ref.c")
_LIB = os.path.join(_HERE, "liboracle_dense.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle next to its source (gcc, AVX2+FMA so fmaf is one instruction)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O3", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC",
               _SRC, "-o", _LIB, "-lm"]
        subprocess.check_call(cmd)
    return _LIB


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = ctypes.CDLL(_LIB)
        lib.oracle_dense_topk.restype = ctypes.c_int
        lib.oracle_dense_topk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int64]
        lib.oracle_dense_scores.restype = ctypes.c_int
        lib.oracle_dense_scores.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_int, ctypes.c_void_p]
        _lib = lib
    return _lib


def dense_topk(Q: np.ndarray, C: np.ndarray, k: int, id_base: int = 0):
    """(scores f32 [nq, k], ids i64 [nq, k]), best first; see dense_topk_ref.c for the definition."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    C = np.ascontiguousarray(C, dtype=np.float32)
    nq, d = Q.shape
    N = C.shape[0]
    assert C.shape[1] == d
    out_s = np.empty((nq, k), np.float32)
    out_i = np.empty((nq, k), np.int64)
    rc = _load().oracle_dense_topk(Q.ctypes.data, nq, C.ctypes.data, N, d, k, out_s.ctypes.data,
                                   out_i.ctypes.data, id_base)
    if rc != 0:
        raise RuntimeError(f"oracle_dense_topk failed: {rc}")
    return out_s, out_i


def dense_scores(Q: np.ndarray, C: np.ndarray) -> np.ndarray:
    """All scores, [N, nq], same arithmetic as dense_topk."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    C = np.ascontiguousarray(C, dtype=np.float32)
    out = np.empty((C.shape[0], Q.shape[0]), np.float32)
    _load().oracle_dense_scores(Q.ctypes.data, Q.shape[0], C.ctypes.data, C.shape[0], Q.shape[1], out.ctypes.data)
    return out


def merge_topk(parts_s: np.ndarray, parts_i: np.ndarray, k: int):
    """Merge [n_parts, nq, k] partial lists by (score desc, id asc); id < 0 marks a fill slot."""
    n_parts, nq, kk = parts_s.shape
    out_s = np.full((nq, k), -np.finfo(np.float32).max, np.float32)
    out_i = np.full((nq, k), -1, np.int64)
    for q in range(nq):
        s = parts_s[:, q, :].reshape(-1)
        i = parts_i[:, q, :].reshape(-1)
        keep = i >= 0
        s, i = s[keep], i[keep]
        order = np.lexsort((i, -s.astype(np.float64)))[:k]
        out_s[q, : len(order)] = s[order]
        out_i[q, : len(order)] = i[order]
    return out_s, out_i
