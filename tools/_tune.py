"""Tools run against the TUNING build of the library (libatlas_hip_tune.so, -DATLAS_TUNING=1): import this FIRST. It points
atlas_amd._lib at that build for this process (ATLAS_HIP_SO) and binds the atlas_tune_* hooks, so that the product classes used by a
tool (HipDistributedIndex, Contriever) run the variant / configuration the tool selects."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ATLAS_HIP_SO", os.path.join(ROOT, "atlas_amd", "lib", "libatlas_hip_tune.so"))

from atlas_amd import _lib  # noqa: E402

L = _lib.lib()
for _name, _args in (("atlas_tune_set_scan_variant", [ctypes.c_int]), ("atlas_tune_set_scan_coop", [ctypes.c_int]), ("atlas_tune_set_scan_fused", [ctypes.c_int]), ("atlas_tune_set_scan_wide", [ctypes.c_int]), ("atlas_tune_set_scan_pair", [ctypes.c_int]), ("atlas_tune_set_scan_gemm", [ctypes.c_int]), ("atlas_tune_set_scan_pool", [ctypes.c_int, ctypes.c_int]), ("atlas_tune_set_gemm_cfg", [ctypes.c_int]),
                     ("atlas_tune_set_gemm_diag", [ctypes.c_int]), ("atlas_tune_set_gemm_stamps", [ctypes.c_void_p]), ("atlas_tune_set_gemm_stamps_nth", [ctypes.c_void_p, ctypes.c_int]),
                     ("atlas_tune_set_merge_stamps", [ctypes.c_void_p]), ("atlas_tune_set_scan_stamps", [ctypes.c_void_p]),
                     ("atlas_tune_set_att_pf", [ctypes.c_int]), ("atlas_tune_set_skip_ln", [ctypes.c_int]), ("atlas_tune_set_att_xmap", [ctypes.c_int])):
    getattr(L, _name).argtypes, getattr(L, _name).restype = _args, None
