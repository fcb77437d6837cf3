"""ctypes front end of oracle/knn_chain.c.  TEST INFRASTRUCTURE ONLY (see knn_oracle.py header)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libknn_chain.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "knn_chain.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libknn_chain.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        i64, p = ctypes.c_int64, ctypes.c_void_p
        L.oracle_sim_chain_f32.argtypes = [p, i64, p, i64, i64, i64, i64, p]
        L.oracle_sim_bf16_f64.argtypes = [p, i64, p, i64, i64, i64, i64, p]
        L.oracle_topk_row.argtypes = [p, i64, ctypes.c_int, p, p]
        for f in (L.oracle_sim_chain_f32, L.oracle_sim_bf16_f64, L.oracle_topk_row):
            f.restype = None
        _lib = L
    return _lib


def sim_chain_f32(q: np.ndarray, bank_rows: np.ndarray) -> np.ndarray:
    """[B, D] x [N, D] -> [B, N], k-ascending float32 fmaf chain (the fp32 kernel's order)."""
    q = np.ascontiguousarray(q, np.float32)
    x = np.ascontiguousarray(bank_rows, np.float32)
    B, D = q.shape
    N = x.shape[0]
    assert x.shape[1] == D
    out = np.empty((B, N), np.float32)
    lib().oracle_sim_chain_f32(q.ctypes.data, D, x.ctypes.data, D, B, N, D, out.ctypes.data)
    return out


def sim_bf16_f64(q_bits: np.ndarray, bank_bits: np.ndarray) -> np.ndarray:
    """bf16 bit patterns in, exact products, float64 accumulate, one rounding to float32."""
    q = np.ascontiguousarray(q_bits, np.uint16)
    x = np.ascontiguousarray(bank_bits, np.uint16)
    B, D = q.shape
    N = x.shape[0]
    out = np.empty((B, N), np.float32)
    lib().oracle_sim_bf16_f64(q.ctypes.data, D, x.ctypes.data, D, B, N, D, out.ctypes.data)
    return out


def topk_rows(sim: np.ndarray, k: int):
    """Row-wise top-k under (value desc, index asc) in O(N*k); for large N."""
    sim = np.ascontiguousarray(sim, np.float32)
    B, N = sim.shape
    v = np.empty((B, k), np.float32)
    i = np.empty((B, k), np.int64)
    L = lib()
    for b in range(B):
        L.oracle_topk_row(sim[b].ctypes.data, N, k, v[b].ctypes.data, i[b].ctypes.data)
    return v, i
