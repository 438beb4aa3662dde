"""ctypes front-end of the CPU oracle (oracle/hm_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py; nothing under ``hyptokenizer_amd/`` may import this module.
All arrays are numpy, fp32 / int32, C-contiguous; the table layout is the reference's
(``[n, d+1]`` row-major, column 0 = time coordinate, ``hyperbolic_merge.py:145-153``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhm_oracle.so")

SIGN_REFERENCE = 0   # literal arithmetic of the reference as shipped
SIGN_LORENTZ = 1     # Minkowski form with the standard sign


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "hm_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libhm_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        f32p, i32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        for name in ("hmo_log1pf", "hmo_acoshf", "hmo_expm1f", "hmo_coshf", "hmo_sinhf"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.hmo_minkowski_u.restype = C.c_float
        L.hmo_minkowski_u.argtypes = [f32p, f32p, C.c_int, C.c_int]
        L.hmo_distance.restype = C.c_float
        L.hmo_distance.argtypes = [f32p, f32p, C.c_int, C.c_float, C.c_int]
        L.hmo_batch_distance.restype = None
        L.hmo_batch_distance.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_int64, C.c_int, C.c_float,
                                         C.c_int, f32p]
        L.hmo_pair_distance.restype = None
        L.hmo_pair_distance.argtypes = [f32p, C.c_int64, C.c_int, i32p, i32p, C.c_int64, C.c_float, C.c_int, f32p]
        L.hmo_row_vs_all.restype = None
        L.hmo_row_vs_all.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_float, C.c_int, f32p]
        L.hmo_log_map.restype = None
        L.hmo_log_map.argtypes = [f32p, f32p, C.c_int, C.c_int, f32p]
        L.hmo_exp_map.restype = None
        L.hmo_exp_map.argtypes = [f32p, f32p, C.c_int, f32p]
        L.hmo_project.restype = None
        L.hmo_project.argtypes = [f32p, C.c_int, C.c_float, f32p]
        L.hmo_midpoint.restype = None
        L.hmo_midpoint.argtypes = [f32p, f32p, C.c_float, C.c_int, C.c_float, C.c_int, f32p]
        L.hmo_midpoint_batch.restype = None
        L.hmo_midpoint_batch.argtypes = [f32p, C.c_int64, C.c_int, i32p, i32p, f32p, C.c_int64, C.c_float,
                                         C.c_int, f32p]
        L.hmo_coherence_distances.restype = None
        L.hmo_coherence_distances.argtypes = [f32p, C.c_int64, C.c_int, i32p, i32p, f32p, i32p, C.c_int64, C.c_int,
                                              C.c_float, C.c_int, f32p]
        L.hmo_project_table.restype = None
        L.hmo_project_table.argtypes = [f32p, C.c_int64, C.c_int, C.c_int64, C.c_float]
        L.hmo_pairwise_count.restype = C.c_int64
        L.hmo_pairwise_count.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int,
                                         C.c_int64, C.c_int64]
        L.hmo_pairwise_candidates.restype = C.c_int64
        L.hmo_pairwise_candidates.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int,
                                              C.c_int64, C.c_int64, C.c_int64, i32p, i32p, f32p]
        for name in ("hmo_pairwise_topk", "hmo_fast_pairwise_topk"):
            fn = getattr(L, name)
            fn.restype = C.c_int64
            fn.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int64,
                           C.c_int64, C.c_int64, f32p, i32p, i32p, i64p]
        L.hmo_num_threads.restype = C.c_int
        L.hmo_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


# ---------------------------------------------------------------------------------------------
# scalar math
# ---------------------------------------------------------------------------------------------
def acosh(a: float) -> float:
    return float(lib().hmo_acoshf(C.c_float(a)))


def log1p(a: float) -> float:
    return float(lib().hmo_log1pf(C.c_float(a)))


def expm1(a: float) -> float:
    return float(lib().hmo_expm1f(C.c_float(a)))


def cosh(a: float) -> float:
    return float(lib().hmo_coshf(C.c_float(a)))


def sinh(a: float) -> float:
    return float(lib().hmo_sinhf(C.c_float(a)))


# ---------------------------------------------------------------------------------------------
# Lorentz primitives (embedding/lorentz_model.py)
# ---------------------------------------------------------------------------------------------
def minkowski_u(x, y, sign_mode: int) -> np.ndarray:
    """Row-wise u (argument of acosh) for x[b, d1], y[b, d1]."""
    x, y = _f32(np.atleast_2d(x)), _f32(np.atleast_2d(y))
    out = np.empty(x.shape[0], np.float32)
    for t in range(x.shape[0]):
        out[t] = lib().hmo_minkowski_u(_p(x[t], C.c_float), _p(y[t], C.c_float), x.shape[1], sign_mode)
    return out


def distance(x, y, c: float, sign_mode: int) -> np.ndarray:
    x, y = _f32(np.atleast_2d(x)), _f32(np.atleast_2d(y))
    out = np.empty(x.shape[0], np.float32)
    for t in range(x.shape[0]):
        out[t] = lib().hmo_distance(_p(x[t], C.c_float), _p(y[t], C.c_float), x.shape[1], c, sign_mode)
    return out


def batch_distance(X, Y, c: float, sign_mode: int) -> np.ndarray:
    X, Y = _f32(X), _f32(Y)
    assert X.shape[1] == Y.shape[1]
    out = np.empty((X.shape[0], Y.shape[0]), np.float32)
    lib().hmo_batch_distance(_p(X, C.c_float), X.shape[0], _p(Y, C.c_float), Y.shape[0], X.shape[1],
                             X.shape[1], c, sign_mode, _p(out, C.c_float))
    return out


def pair_distance(X, I, J, c: float, sign_mode: int) -> np.ndarray:
    X, I, J = _f32(X), _i32(I), _i32(J)
    out = np.empty(I.shape[0], np.float32)
    lib().hmo_pair_distance(_p(X, C.c_float), X.shape[1], X.shape[1], _p(I, C.c_int32), _p(J, C.c_int32),
                            I.shape[0], c, sign_mode, _p(out, C.c_float))
    return out


def row_vs_all(X, n: int, row: int, c: float, sign_mode: int) -> np.ndarray:
    X = _f32(X)
    out = np.empty(n, np.float32)
    lib().hmo_row_vs_all(_p(X, C.c_float), n, X.shape[1], X.shape[1], row, c, sign_mode, _p(out, C.c_float))
    return out


def log_map(x, y, sign_mode: int) -> np.ndarray:
    x, y = _f32(np.atleast_2d(x)), _f32(np.atleast_2d(y))
    out = np.empty_like(x)
    for t in range(x.shape[0]):
        lib().hmo_log_map(_p(x[t], C.c_float), _p(y[t], C.c_float), x.shape[1], sign_mode, _p(out[t], C.c_float))
    return out


def exp_map(x, v) -> np.ndarray:
    x, v = _f32(np.atleast_2d(x)), _f32(np.atleast_2d(v))
    out = np.empty_like(x)
    for t in range(x.shape[0]):
        lib().hmo_exp_map(_p(x[t], C.c_float), _p(v[t], C.c_float), x.shape[1], _p(out[t], C.c_float))
    return out


def project(x, c: float) -> np.ndarray:
    x = _f32(np.atleast_2d(x))
    out = np.empty_like(x)
    for t in range(x.shape[0]):
        lib().hmo_project(_p(x[t], C.c_float), x.shape[1], c, _p(out[t], C.c_float))
    return out


def midpoint_batch(X, I, J, W, c: float, sign_mode: int) -> np.ndarray:
    X, I, J, W = _f32(X), _i32(I), _i32(J), _f32(W)
    out = np.empty((I.shape[0], X.shape[1]), np.float32)
    lib().hmo_midpoint_batch(_p(X, C.c_float), X.shape[1], X.shape[1], _p(I, C.c_int32), _p(J, C.c_int32),
                             _p(W, C.c_float), I.shape[0], c, sign_mode, _p(out, C.c_float))
    return out


def coherence_distances(X, I, J, W, S, c: float, sign_mode: int) -> np.ndarray:
    """enhanced_fast_hyperbolic_merge.py:308-333: distances from the un-projected simulated midpoint of
    (I[t], J[t]) to the sampled rows S[t, :].  -> [b, ns] fp32."""
    X, I, J, W = _f32(X), _i32(I), _i32(J), _f32(W)
    S = _i32(S).reshape(I.shape[0], -1)
    out = np.empty(S.shape, np.float32)
    if S.size:
        lib().hmo_coherence_distances(_p(X, C.c_float), X.shape[1], X.shape[1], _p(I, C.c_int32), _p(J, C.c_int32),
                                      _p(W, C.c_float), _p(S, C.c_int32), I.shape[0], S.shape[1], c, sign_mode,
                                      _p(out, C.c_float))
    return out


def project_table(X: np.ndarray, n: int, c: float) -> None:
    """enhanced...:784-792: project rows [0, n) of the fp32 C-contiguous table X in place."""
    assert X.dtype == np.float32 and X.flags["C_CONTIGUOUS"]
    lib().hmo_project_table(_p(X, C.c_float), X.shape[1], X.shape[1], n, c)


# ---------------------------------------------------------------------------------------------
# candidate search (tokenizer/hyperbolic_merge.py:192-291, fast_hyperbolic_merge.py:253-376)
# ---------------------------------------------------------------------------------------------
def pairwise_count(X, n: int, c: float, thr: float, sign_mode: int, row_begin: int = 0, row_end: int = -1) -> int:
    X = _f32(X)
    if row_end < 0:
        row_end = n
    return int(lib().hmo_pairwise_count(_p(X, C.c_float), n, X.shape[1], X.shape[1], c, thr, sign_mode,
                                        row_begin, row_end))


def pairwise_candidates(X, n: int, c: float, thr: float, sign_mode: int, cap: int = 1 << 22,
                        row_begin: int = 0, row_end: int = -1):
    """Row-major (i, j, d) list, as `_find_merge_candidates` returns it.  -> (i, j, d, total)."""
    X = _f32(X)
    if row_end < 0:
        row_end = n
    oi, oj, od = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
    total = int(lib().hmo_pairwise_candidates(_p(X, C.c_float), n, X.shape[1], X.shape[1], c, thr, sign_mode,
                                              row_begin, row_end, cap, _p(oi, C.c_int32), _p(oj, C.c_int32),
                                              _p(od, C.c_float)))
    m = min(total, cap)
    return oi[:m].copy(), oj[:m].copy(), od[:m].copy(), total


def pairwise_topk(X, n: int, c: float, thr: float, sign_mode: int, k: int, row_begin: int = 0,
                  row_end: int = -1, fast: bool = False):
    """k smallest candidates in the reference's (d, i, j) order.  -> (d, i, j, count)."""
    X = _f32(X)
    if row_end < 0:
        row_end = n
    od, oi, oj = np.empty(k, np.float32), np.empty(k, np.int32), np.empty(k, np.int32)
    cnt = C.c_int64(0)
    fn = lib().hmo_fast_pairwise_topk if fast else lib().hmo_pairwise_topk
    m = int(fn(_p(X, C.c_float), n, X.shape[1], X.shape[1], c, thr, sign_mode, row_begin, row_end, k,
               _p(od, C.c_float), _p(oi, C.c_int32), _p(oj, C.c_int32), C.byref(cnt)))
    return od[:m].copy(), oi[:m].copy(), oj[:m].copy(), int(cnt.value)


def num_threads() -> int:
    return int(lib().hmo_num_threads())


def set_num_threads(n: int) -> None:
    lib().hmo_set_num_threads(n)


# --------------------------------------------------------------------------------------------------
# tokenize / encode (tokenizer/hyperbolic_merge.py:414-459): plain Python, as the reference's is
# --------------------------------------------------------------------------------------------------
def merge_rules(merge_history) -> dict:
    """{(old1, old2): new}; a later entry for the same pair replaces the earlier one (:425-428)."""
    rules = {}
    for old1, old2, new in merge_history:
        rules[(old1, old2)] = new
    return rules


def tokenize(rules: dict, text: str, count_passes: bool = False):
    """Repeated left-to-right passes; a hit rewrites position i, removes i + 1 and stays at i (:430-446)."""
    tokens = list(text)
    passes = 0
    changed = True
    while changed:
        changed = False
        passes += 1
        i = 0
        while i < len(tokens) - 1:
            hit = rules.get((tokens[i], tokens[i + 1]))
            if hit is not None:
                tokens[i] = hit
                del tokens[i + 1]
                changed = True
            else:
                i += 1
    return (tokens, passes) if count_passes else tokens


def encode(rules: dict, token2idx: dict, text: str) -> list:
    unk = token2idx.get("<unk>", 3)        # :459
    return [token2idx.get(t, unk) for t in tokenize(rules, text)]
