"""The C-ABI library builds for gfx950, loads, and exports every symbol include/atlas_hip.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    from atlas_amd import build

    return build.build_hip()


@pytest.fixture(scope="module")
def tune_so():
    """the -DATLAS_TUNING=1 build of the same sources: the host-side test hooks (plan word, pass geometry) live there, not in the product"""
    from atlas_amd import build

    return build.build_hip(tuning=True)


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(atlas_[a-z0-9_]+)\s*\(", hdr))


def test_header_symbols_exported(so):
    declared = _declared("atlas_hip.h")
    assert len(declared) >= 10
    L = ctypes.CDLL(so)
    missing = [s for s in sorted(declared | _declared("atlas_hip_experimental.h")) if not hasattr(L, s)]
    assert not missing, missing
    from atlas_amd import _lib

    assert set(_lib.SYMBOLS) == declared
    assert set(_lib.EXPERIMENTAL_SYMBOLS) == _declared("atlas_hip_experimental.h")


def test_product_exports_are_exactly_the_headers(so):
    """the header's "no hooks in the product" is enforced -- every dynamic `atlas_*` symbol libatlas_hip.so defines is
    declared in include/atlas_hip.h (the product interface) or include/atlas_hip_experimental.h (VERDICT r05 next #1d: the peer exchange, which
    has never run across two devices, is kept out of the product header) and vice versa; test hooks and tuning knobs exist in
    libatlas_hip_tune.so only"""
    import subprocess

    product, experimental = _declared("atlas_hip.h"), _declared("atlas_hip_experimental.h")
    assert not product & experimental
    assert not [s for s in product if "xchg" in s] and all("xchg" in s for s in experimental)
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in ("T", "t", "D", "B", "R", "W")}
    atlas = {s for s in exported if s.startswith("atlas_")}
    assert atlas == product | experimental, (sorted(atlas - product - experimental), sorted((product | experimental) - atlas))
    assert not [s for s in exported if "tune" in s or "test" in s or "dbg" in s], exported


def test_abi_version_and_bad_args(so):
    L = ctypes.CDLL(so)
    L.atlas_abi_version.restype = ctypes.c_int
    assert L.atlas_abi_version() == 9
    L.atlas_build_info.restype = ctypes.c_char_p
    assert b"gfx950" in L.atlas_build_info()
    # argument validation happens before any HIP call
    L.atlas_scan_topk.restype = ctypes.c_int
    assert L.atlas_scan_topk(None, 0, None, ctypes.c_int64(10), 1, 768, 4, ctypes.c_float(1.0), None, None, None, None,
                             ctypes.c_size_t(0), None) == -1
    L.atlas_merge_packed.restype = ctypes.c_int
    assert L.atlas_merge_packed(None, 1, 1, 1, None, None) == -1


def _plan(L, N, B, k=40, cus=256):
    from atlas_amd import _lib

    w = L.atlas_test_plan_word(ctypes.c_int64(N), B, k, cus)
    assert w >= 0, (N, B, w)
    return _lib.decode_plan(w)


def test_pass_planner_without_a_device(tune_so):
    """the passes a batch is made of (atlas_hip.hip::plan_batch, pure host arithmetic behind a test hook; ATLAS_ST_PLAN reports the same word from
    a real call): one streaming pass up to 64 / 96 queries, GEMM-shaped passes of 128 / 192 / 256 / 384 / 512 / 1024 queries above that -- from 65
    queries on shards of >= 4M rows --, the streaming passes of round 3 on shards below 65 536 rows, and never more slab reads than queries / 64"""
    L = ctypes.CDLL(tune_so)
    L.atlas_test_plan_word.restype = ctypes.c_int
    one = lambda **kw: dict({"passes_64": 0, "passes_96": 0, "pairs_64": 0, "pairs_96": 0, "gemm_passes": 0}, **kw)
    for N in (10_000, 1_000_000, 4_000_000, 32_000_000):
        for B in (1, 7, 64):
            assert _plan(L, N, B) == one(passes_64=1), (N, B)
    assert _plan(L, 3_999_999, 96) == one(passes_96=1) and _plan(L, 1_000_000, 65) == one(passes_96=1)
    assert _plan(L, 4_000_000, 96) == one(gemm_passes=1) and _plan(L, 32_000_000, 65) == one(gemm_passes=1)      # GS_SMALL_BATCH_MIN_ROWS
    for B in (97, 128, 129, 192, 193, 256, 257, 384, 385, 512, 1024):
        for N in (65_536, 1_000_000, 4_000_000, 32_000_000):
            assert _plan(L, N, B) == one(gemm_passes=1), (N, B)                                                   # one slab read whatever the width
    assert _plan(L, 4_000_000, 513) == one(gemm_passes=2)                                                         # 384 + 192 (3.62) beats 512 + 64 (3.69) and 1024 (5.02)
    assert _plan(L, 4_000_000, 640) == one(gemm_passes=2) and _plan(L, 4_000_000, 768) == one(gemm_passes=2)     # 512 + 128, 512 + 256
    assert _plan(L, 4_000_000, 1100) == one(gemm_passes=2) and _plan(L, 4_000_000, 2048) == one(gemm_passes=2)
    p = _plan(L, 4_000_000, 320)                                                                                  # 384-wide, or 256 + 64
    assert p in (one(gemm_passes=1), one(gemm_passes=1, passes_64=1)), p
    # below 65 536 rows no GEMM-shaped pass: round 3's streaming passes (single, wide, paired)
    p = _plan(L, 60_000, 512)
    assert p["gemm_passes"] == 0 and sum(p.values()) >= 3, p
    # a smaller device (a partition of 64 CUs): plans exist, nothing asks for more workgroups than CUs
    assert sum(_plan(L, 4_000_000, 512, cus=64).values()) >= 1
    # the counters of the plan word saturate at their field widths instead of carrying into the next field (a huge batch on a shard
    # the GEMM-shaped pass does not take: thousands of streaming passes)
    for B in (64 * 300, 96 * 700 + 5):
        p = _plan(L, 60_000, B)
        assert p["gemm_passes"] == 0 and p["passes_64"] <= 255 and p["passes_96"] <= 255 and p["pairs_64"] <= 15 and p["pairs_96"] <= 15, p
        assert max(p["passes_64"], p["passes_96"]) == 255 or max(p["pairs_64"], p["pairs_96"]) == 15, p
    # k beyond the fast path / a negative row count: the error codes of atlas_scan_topk
    assert L.atlas_test_plan_word(ctypes.c_int64(1000), 64, 300, 256) == -3 and L.atlas_test_plan_word(ctypes.c_int64(-1), 64, 40, 256) == -1


def test_gemm_shaped_pass_geometry_without_a_device(tune_so):
    """make_gplan (launch geometry + workspace layout of one GEMM-shaped pass) over random shard sizes, pass sizes and CU counts: the column tiles
    hold the queries, the row ranges tile the slab in whole 256-row tiles below the 24-bit row field of a candidate entry, the sample's tiles lie
    inside the slab, the workspace regions are aligned, ordered and big enough"""
    import numpy as np

    L = ctypes.CDLL(tune_so)
    out = (ctypes.c_int64 * 17)()
    rng = np.random.default_rng(5)
    seen_ok = 0
    for _ in range(3000):
        N = int(rng.choice([rng.integers(1, 70_000), rng.integers(65_536, 300_000), rng.integers(300_000, 40_000_000), rng.integers(40_000_000, 4_000_000_000)]))
        nq = int(rng.integers(1, 1100))
        cus = int(rng.choice([8, 32, 64, 104, 128, 256, 304]))
        L.atlas_test_gplan(ctypes.c_int64(N), nq, cus, out)
        ok, G, ncol, cw, ldq, rpr, s_tiles, s_stride, nmax, gcap, o_q16, o_th, o_cnt, o_st, o_sm, o_li, total = (int(x) for x in out)
        if not ok:
            assert nq > 1024 or N // 256 < 256 or G < 8 * ncol or (N + 255) // 256 / max(1, G // max(1, ncol)) * 256 >= (1 << 24) - 256, (N, nq, cus)
            continue
        seen_ok += 1
        assert cw in (128, 192, 256) and ncol in (1, 2, 4) and ldq == ncol * cw >= nq and (ldq < 2 * nq or ldq == 128), (nq, cw, ncol)
        assert G % (8 * ncol) == 0 and 8 * ncol <= G <= cus and G <= 1024
        nranges = G // ncol
        # (trailing ranges may be empty: their workgroups return at once) the ranges cover the slab with the fewest whole tiles per range
        assert rpr % 256 == 0 and rpr < (1 << 24) and nranges * rpr >= N and (rpr // 256 - 1) * nranges < (N + 255) // 256
        assert 128 <= s_tiles <= 2048 and nmax == 16 * s_tiles and s_stride % 256 == 0 and s_stride >= 256
        assert (s_tiles - 1) * s_stride + 256 <= N                                       # every sampled tile is a whole tile inside the slab
        assert gcap == 32768
        offs = [o_q16, o_th, o_cnt, o_st, o_sm, o_li, total]
        assert all(o % 256 == 0 for o in offs[:-1]) and all(a < b for a, b in zip(offs, offs[1:]))
        assert o_th - o_q16 >= ldq * 768 * 2 and o_cnt - o_th >= ldq * 4 and o_st - o_cnt >= ldq * 4 and o_sm - o_st >= G * 8 * 2
        assert o_li - o_sm >= nmax * ldq * 4 and total - o_li == ldq * gcap * 8
    assert seen_ok > 1000


def test_gfx950_code_object(so):
    """the shared library embeds a gfx950 code object and nothing else"""
    data = open(so, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in data


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under atlas_amd/ may reference it"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "atlas_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from atlas_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "HIP_SO", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AtlasHipError, match="no CPU fallback"):
        _lib.lib()


def test_product_library_carries_no_tuning_only_kernels(so):
    """every kernel in libatlas_hip.so is one its own dispatch can launch -- the A/B references of the GEMM configurations
    (two-workgroup kernel, 256 x 256 single-phase, 64 x 64 single-stage, weights-in-registers, the 16-bit instantiations of the LDS-epilogue
    ping-pong kernel), the scan variants and the spin kernel exist in the tuning build only"""
    import subprocess

    names = subprocess.run(["nm", "-C", so], capture_output=True, text=True, check=True).stdout
    stubs = [ln.split("__device_stub__", 1)[1] for ln in names.splitlines() if "__device_stub__" in ln]
    assert 40 <= len(stubs) <= 90, len(stubs)             # (round 5: 82 -- six row-major-V attention instantiations in, the two V^T-epilogue GEMMs out)
    banned = ("gemm_co_kernel", "gemm_wr_kernel", "atlas_spin_kernel", "gemm_pp_kernel<F16", "gemm_pp_kernel<BF16", ", 256, 256, 2, 4>", ", 64, 64, 2, 2>",
              "scan_kernel<8,", "scan_kernel<12,", "scan_kernel<16, 2,")
    bad = [s for s in stubs if any(b in s for b in banned)]
    assert not bad, bad
