"""ctypes front-end of oracle/oracle.c (see that file's header for what is restated and why).

TEST INFRASTRUCTURE ONLY — never imported by atlas_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u16p, i64p, f32p = (ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_float))
        L.oracle_f16_to_f64.restype = ctypes.c_double
        L.oracle_f16_to_f64.argtypes = [ctypes.c_uint16]
        L.oracle_f64_to_f16.restype = ctypes.c_uint16
        L.oracle_f64_to_f16.argtypes = [ctypes.c_double]
        L.oracle_f32_to_f16.restype = ctypes.c_uint16
        L.oracle_f32_to_f16.argtypes = [ctypes.c_float]
        L.oracle_bf16_to_f16.restype = ctypes.c_uint16
        L.oracle_bf16_to_f16.argtypes = [ctypes.c_uint16]
        L.oracle_key16.restype = ctypes.c_uint16
        L.oracle_key16.argtypes = [ctypes.c_uint16]
        L.oracle_f32_to_f16_array.restype = None
        L.oracle_f32_to_f16_array.argtypes = [f32p, ctypes.c_int64, u16p]
        L.oracle_exact_dot.restype = ctypes.c_double
        L.oracle_exact_dot.argtypes = [u16p, u16p, ctypes.c_int]
        L.oracle_score_row.restype = None
        L.oracle_score_row.argtypes = [u16p, u16p, ctypes.c_int64, ctypes.c_int, u16p]
        L.oracle_topk_row.restype = None
        L.oracle_topk_row.argtypes = [u16p, ctypes.c_int64, ctypes.c_int, u16p, i64p]
        L.oracle_search.restype = None
        L.oracle_search.argtypes = [u16p, u16p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    u16p, i64p, u16p]
        L.oracle_merge.restype = None
        L.oracle_merge.argtypes = [u16p, i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, u16p, i64p]
        L.oracle_pool.restype = None
        L.oracle_pool.argtypes = [u16p, i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, u16p]
        _lib = L
    return _lib


def _u16(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        a = a.view(np.uint16)
    assert a.dtype == np.uint16, a.dtype
    return a


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def f32_to_f16(x) -> np.ndarray:
    """fp32 -> fp16 RNE (== torch `.half()`), returns float16 array."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().oracle_f32_to_f16_array(_p(x, ctypes.c_float), x.size, _p(out, ctypes.c_uint16))
    return out.view(np.float16)


def search(q_f16, slab_f16, k, return_full=False):
    """Canonical exact-MIPS top-k. q [B,d] fp16, slab [N,d] fp16 -> (scores fp16 [B,k], idx int64 [B,k])."""
    q = _u16(q_f16)
    s = _u16(slab_f16)
    B, d = q.shape
    N = s.shape[0]
    assert s.shape[1] == d
    out_s = np.empty((B, k), dtype=np.uint16)
    out_i = np.empty((B, k), dtype=np.int64)
    full = np.empty((B, N), dtype=np.uint16) if return_full else None
    lib().oracle_search(_p(q, ctypes.c_uint16), _p(s, ctypes.c_uint16), N, B, d, k, _p(out_s, ctypes.c_uint16),
                        _p(out_i, ctypes.c_int64), _p(full, ctypes.c_uint16) if return_full else None)
    if return_full:
        return out_s.view(np.float16), out_i, full.view(np.float16)
    return out_s.view(np.float16), out_i


def score_row(q_f16, slab_f16) -> np.ndarray:
    """canonical fp16 scores of ONE query against every row of `slab_f16` [N, d] (OpenMP over rows); lets a caller stream a
    slab that does not fit in host memory through the oracle chunk by chunk and finish with topk_row"""
    q = _u16(q_f16).reshape(-1)
    s = _u16(slab_f16)
    N, d = s.shape
    assert q.shape[0] == d
    out = np.empty(N, dtype=np.uint16)
    lib().oracle_score_row(_p(q, ctypes.c_uint16), _p(s, ctypes.c_uint16), N, d, _p(out, ctypes.c_uint16))
    return out.view(np.float16)


def topk_row(scores_f16, k):
    s = _u16(scores_f16)
    out_s = np.empty(k, dtype=np.uint16)
    out_i = np.empty(k, dtype=np.int64)
    lib().oracle_topk_row(_p(s, ctypes.c_uint16), s.shape[0], k, _p(out_s, ctypes.c_uint16), _p(out_i, ctypes.c_int64))
    return out_s.view(np.float16), out_i


def merge(scores_f16, gids):
    """scores [W,B,k] fp16, gids [W,B,k] int64 (-1 = padding) -> ([B,k] fp16, [B,k] int64)."""
    s = _u16(scores_f16)
    g = np.ascontiguousarray(gids, dtype=np.int64)
    W, B, k = s.shape
    out_s = np.empty((B, k), dtype=np.uint16)
    out_g = np.empty((B, k), dtype=np.int64)
    lib().oracle_merge(_p(s, ctypes.c_uint16), _p(g, ctypes.c_int64), W, B, k, _p(out_s, ctypes.c_uint16),
                       _p(out_g, ctypes.c_int64))
    return out_s.view(np.float16), out_g


def pool(hidden_f16, mask):
    h = _u16(hidden_f16)
    m = np.ascontiguousarray(mask, dtype=np.int64)
    n, L, d = h.shape
    out = np.empty((n, d), dtype=np.uint16)
    lib().oracle_pool(_p(h, ctypes.c_uint16), _p(m, ctypes.c_int64), n, L, d, _p(out, ctypes.c_uint16))
    return out.view(np.float16)


def exact_dot(q_f16, p_f16) -> float:
    q = _u16(q_f16)
    p = _u16(p_f16)
    return lib().oracle_exact_dot(_p(q, ctypes.c_uint16), _p(p, ctypes.c_uint16), q.shape[0])
