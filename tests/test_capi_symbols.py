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


def test_header_symbols_exported(so):
    hdr = open(os.path.join(ROOT, "include", "atlas_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(atlas_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 10
    L = ctypes.CDLL(so)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    from atlas_amd import _lib

    assert set(_lib.SYMBOLS) == declared


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
    assert 40 <= len(stubs) <= 80, len(stubs)
    banned = ("gemm_co_kernel", "gemm_wr_kernel", "atlas_spin_kernel", "gemm_pp_kernel<F16", "gemm_pp_kernel<BF16", ", 256, 256, 2, 4>", ", 64, 64, 2, 2>",
              "scan_kernel<8,", "scan_kernel<12,", "scan_kernel<16, 2,")
    bad = [s for s in stubs if any(b in s for b in banned)]
    assert not bad, bad
