"""The arithmetic helpers the HIP kernels use (atlas_amd/csrc/common.h, compiled here with g++) against the
independent restatement in oracle/oracle.c and against numpy. No GPU needed: same source the device runs."""
import ctypes

import numpy as np
import pytest

import synth


@pytest.fixture(scope="module")
def H():
    from atlas_amd import build

    L = ctypes.CDLL(build.build_host())
    L.h_f16_to_f64.restype = ctypes.c_double
    L.h_f16_to_f64.argtypes = [ctypes.c_uint16]
    L.h_f64_to_f16.restype = ctypes.c_uint16
    L.h_f64_to_f16.argtypes = [ctypes.c_double]
    L.h_f64_to_f16_rto.restype = ctypes.c_uint16
    L.h_f64_to_f16_rto.argtypes = [ctypes.c_double]
    L.h_f32_to_f16.restype = ctypes.c_uint16
    L.h_f32_to_f16.argtypes = [ctypes.c_float]
    L.h_bf16_to_f16.restype = ctypes.c_uint16
    L.h_bf16_to_f16.argtypes = [ctypes.c_uint16]
    L.h_f16_order_key.restype = ctypes.c_uint16
    L.h_f16_order_key.argtypes = [ctypes.c_uint16]
    L.h_f16_from_order_key.restype = ctypes.c_uint16
    L.h_f16_from_order_key.argtypes = [ctypes.c_uint16]
    L.h_f32_order_key.restype = ctypes.c_uint32
    L.h_f32_order_key.argtypes = [ctypes.c_float]
    L.h_f32_from_order_key.restype = ctypes.c_float
    L.h_f32_from_order_key.argtypes = [ctypes.c_uint32]
    L.h_local_key.restype = ctypes.c_uint64
    L.h_local_key.argtypes = [ctypes.c_uint16, ctypes.c_uint32]
    L.h_pack_candidate.restype = ctypes.c_uint64
    L.h_pack_candidate.argtypes = [ctypes.c_uint16, ctypes.c_uint64]
    L.h_exact_dot.restype = ctypes.c_double
    L.h_exact_dot.argtypes = [ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16), ctypes.c_int]
    L.h_ulp16_at.restype = ctypes.c_float
    L.h_ulp16_at.argtypes = [ctypes.c_float]
    L.h_prune_threshold.restype = ctypes.c_float
    L.h_prune_threshold.argtypes = [ctypes.c_float, ctypes.c_float]
    return L


def test_f16_to_f64_all_values(H, oracle_mod):
    O = oracle_mod.lib()
    for h in range(65536):
        a, b = H.h_f16_to_f64(h), O.oracle_f16_to_f64(h)
        assert a == b or (np.isnan(a) and np.isnan(b)), h


def test_f64_to_f16_rounding(H, oracle_mod):
    O = oracle_mod.lib()
    rng = np.random.default_rng(5)
    xs = list(rng.standard_normal(4000) * np.exp(rng.uniform(-22, 13, 4000)))
    # every midpoint between adjacent fp16 values, and its neighbours (ties-to-even, carries, subnormals, inf)
    allh = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = (allh[:-1] + allh[1:]) / 2
    for m in mids[:: 7]:
        xs += [m, np.nextafter(m, 0), np.nextafter(m, np.inf), -m]
    xs += [0.0, -0.0, 65504.0, 65519.99, 65520.0, 65536.0, 1e300, -1e300, np.inf, -np.inf, 2.0 ** -25, 2.0 ** -25 * 1.0000001,
           2.0 ** -24, 5e-324]
    for x in xs:
        a, b = H.h_f64_to_f16(float(x)), O.oracle_f64_to_f16(float(x))
        n = int(np.float64(x).astype(np.float16).view(np.uint16))
        assert a == b == n == H.h_f64_to_f16_rto(float(x)), (x, a, b, n, H.h_f64_to_f16_rto(float(x)))
    assert H.h_f64_to_f16(float("nan")) & 0x7C00 == 0x7C00


def test_f32_and_bf16_to_f16(H, oracle_mod):
    O = oracle_mod.lib()
    f = synth.normal_f32(64, 768, 9, 3.0).ravel()[:20000]
    for x in f:
        assert H.h_f32_to_f16(float(x)) == O.oracle_f32_to_f16(float(x)) == int(np.float32(x).astype(np.float16).view(np.uint16))
    for b in range(0, 65536, 13):
        assert H.h_bf16_to_f16(b) == O.oracle_bf16_to_f16(b)


def test_order_keys(H, oracle_mod):
    O = oracle_mod.lib()
    vals = np.arange(65536, dtype=np.uint16)
    fin = vals[(vals & 0x7C00) != 0x7C00]          # finite
    keys = np.array([H.h_f16_order_key(int(h)) for h in fin], dtype=np.int64)
    assert all(H.h_f16_order_key(int(h)) == O.oracle_key16(int(h)) for h in fin[::5])
    order = np.argsort(keys, kind="stable")
    f = fin.view(np.float16).astype(np.float64)[order]
    assert np.all(np.diff(f) >= 0)                                          # monotone
    assert H.h_f16_order_key(0x8000) == H.h_f16_order_key(0x0000)            # -0 == +0
    for h in fin[::3]:
        if h != 0x8000:
            assert H.h_f16_from_order_key(H.h_f16_order_key(int(h))) == h
    x = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 2.0, np.inf], dtype=np.float32)
    k = [H.h_f32_order_key(float(v)) for v in x]
    assert k == sorted(k)
    for v in x:
        assert H.h_f32_from_order_key(H.h_f32_order_key(float(v))) == v
    # canonical order: score desc, then row asc
    assert H.h_local_key(0x4000, 5) > H.h_local_key(0x4000, 6) > H.h_local_key(0x3FFF, 0)
    assert H.h_pack_candidate(0x4000, 5) > H.h_pack_candidate(0x4000, 6) > H.h_pack_candidate(0x3FFF, 0) > 0


def test_exact_dot_matches_oracle_bitwise(H, oracle_mod):
    rng = np.random.default_rng(1)
    for d in (768, 8, 7, 1, 100, 1023):
        for _ in range(20):
            q = (rng.standard_normal(d) * np.exp(rng.uniform(-8, 4, d))).astype(np.float16).view(np.uint16)
            p = (rng.standard_normal(d) * np.exp(rng.uniform(-8, 4, d))).astype(np.float16).view(np.uint16)
            a = H.h_exact_dot(q.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), p.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), d)
            b = oracle_mod.exact_dot(q, p)
            assert a == b, (d, a, b)


def test_prune_threshold_is_safe(H):
    """Brute-force the claim behind the pruning margin: if s~ <= theta(T, eps) then for ANY exact scores within
    eps of (s~, T) the fp16 score of the pruned row is strictly below the fp16 score implied by T."""
    rng = np.random.default_rng(2)
    f16 = lambda x: np.float64(x).astype(np.float16).astype(np.float64)   # noqa: E731
    for _ in range(4000):
        T = float(np.float32(rng.standard_normal() * np.exp(rng.uniform(-16, 11))))
        eps = float(np.float32(abs(T) * np.exp(rng.uniform(-14, -3)) + np.exp(rng.uniform(-30, -10))))
        th = H.h_prune_threshold(T, eps)
        assert th < T
        if th == float("-inf"):
            continue                                 # nothing finite is pruned (k-th best already rounds to -inf)
        s_tilde = float(np.float32(th))              # the largest value that is still pruned
        worst_pruned = f16(s_tilde + eps)            # its largest possible fp16 score
        worst_kth = f16(min(T, 65520.0) - eps)       # the smallest possible fp16 score of the k-th best
        assert worst_pruned < worst_kth, (T, eps, th, worst_pruned, worst_kth)
    # boundaries: just below a negative power of two, around zero, overflow
    for T in (-2.0, -2.0000002, -1.9999999, -6.1e-5, 0.0, 6.1e-5, 65519.0, 70000.0, 1e9):
        for eps in (0.0, 1e-6, 1e-3):
            th = H.h_prune_threshold(T, eps)
            assert f16(np.float32(th) + eps) < f16(min(T, 65520.0) - eps), (T, eps)
    assert H.h_prune_threshold(float("-inf"), 0.1) == float("-inf")
    assert H.h_prune_threshold(float("nan"), 0.1) == float("-inf")


def test_gelu_poly(H):
    """the encoder's GELU (common.h::gelu_erf_poly, the same source the kernels compile) over EVERY finite fp16 input:
    its fp16-rounded result is within 1 ulp of the correctly rounded exact-erf GELU, and at least as often exactly right as
    the reference's fp32 formula 0.5*v*(1+erf(v/sqrt2)) evaluated with an ideal fp32 erf"""
    import ctypes
    from scipy import special

    H.h_gelu_erf_poly.restype = ctypes.c_float
    H.h_gelu_erf_poly.argtypes = [ctypes.c_float]
    v = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    v = v[np.isfinite(v)]
    got = np.array([H.h_gelu_erf_poly(float(x)) for x in v], dtype=np.float32)
    exact = 0.5 * v.astype(np.float64) * (1 + special.erf(v.astype(np.float64) / np.sqrt(2)))
    erf32 = special.erf((v * np.float32(0.70710678118654752)).astype(np.float64)).astype(np.float32)
    formula = ((np.float32(0.5) * v) * (np.float32(1) + erf32)).astype(np.float32)

    def ordinal(a):
        i = a.astype(np.float16).view(np.int16).astype(np.int32)
        return np.where(i < 0, -(i & 0x7FFF), i)

    d_poly = np.abs(ordinal(got) - ordinal(exact))
    d_formula = np.abs(ordinal(formula) - ordinal(exact))
    print("gelu fp16: poly wrong on", int((d_poly > 0).sum()), "max", int(d_poly.max()), "ulp; fp32 formula wrong on", int((d_formula > 0).sum()),
          "max", int(d_formula.max()), "ulp; max |poly - exact| =", float(np.abs(got - exact)[np.abs(v) <= 8].max()))
    assert d_poly.max() <= 1 and (d_poly > 0).sum() <= (d_formula > 0).sum()
    assert np.abs(got - exact)[np.abs(v) <= 8].max() <= 4e-7
    assert np.isnan(H.h_gelu_erf_poly(float("nan"))) and H.h_gelu_erf_poly(0.0) == 0.0
    assert H.h_gelu_erf_poly(-65504.0) == 0.0 and H.h_gelu_erf_poly(65504.0) == 65504.0
    assert H.h_gelu_erf_poly(-3.0e38) == 0.0 and H.h_gelu_erf_poly(3.0e38) == np.float32(3.0e38)     # bf16 range
