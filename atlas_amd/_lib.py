"""ctypes binding of libatlas_hip.so (the C-ABI declared in include/atlas_hip.h).

No fallback: `lib()` raises if the library was not built (`python -m atlas_amd.build`).
torch must be imported before the library is loaded so that the HIP runtime torch ships
(SONAME libamdhip64.so.7) is the one already resident; the extension then binds to it and
shares torch's streams and device pointers.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("ATLAS_HIP_SO") or os.path.join(HERE, "lib", "libatlas_hip.so")   # ATLAS_HIP_SO: A/B runs of two builds (dev)
TUNE_SO = os.path.join(HERE, "lib", "libatlas_hip_tune.so")   # the -DATLAS_TUNING=1 build: lib(tuning=True), tools/ and configuration tests only

ABI_VERSION = 9
DT_F16, DT_F32, DT_BF16 = 0, 1, 2
STATUS_HEADER = 8
SCAN_TRUST_PMAX = 1          # ATLAS_SCAN_TRUST_PMAX
ST_FLAGS, ST_PMAX_BITS, ST_N_FALLBACK, ST_N_CANDIDATES, ST_N_RESCORED, ST_MAXERR_BITS, ST_PLAN = 0, 1, 2, 3, 4, 5, 6


def decode_plan(word: int) -> dict:
    """ATLAS_ST_PLAN -> the slab passes a search was made of (include/atlas_hip.h)"""
    word = int(word) & 0xFFFFFFFF
    return {"passes_64": word & 0xFF, "passes_96": (word >> 8) & 0xFF, "pairs_64": (word >> 16) & 0xF, "pairs_96": (word >> 20) & 0xF,
            "gemm_passes": (word >> 24) & 0xFF}
F_PMAX_VIOLATION, F_FALLBACK, F_EPS_VIOLATION = 1, 2, 4
E_BADARG, E_WORKSPACE, E_UNSUPPORTED = -1, -2, -3
D_FAST, K_FAST_MAX, K_EXACT_MAX = 768, 256, 2048

SYMBOLS = [
    "atlas_abi_version", "atlas_build_info",
    "atlas_scan_topk_workspace_bytes", "atlas_scan_topk", "atlas_scan_topk_ex", "atlas_scan_topk_flags", "atlas_scan_topk_pack",
    "atlas_exact_topk_workspace_bytes", "atlas_exact_topk",
    "atlas_pack_candidates", "atlas_merge_packed",
    "atlas_pool_write", "atlas_slab_pmax",
    "atlas_contriever_workspace_bytes", "atlas_contriever_embed", "atlas_contriever_embed_rows",
]
# include/atlas_hip_experimental.h: exported, NOT part of the product interface (the peer exchange has never run across two devices)
EXPERIMENTAL_SYMBOLS = [
    "atlas_xchg_bytes", "atlas_xchg_create", "atlas_xchg_open", "atlas_xchg_close", "atlas_xchg_destroy", "atlas_xchg_push", "atlas_xchg_merge",
]

BERT_MAX_LAYERS = 24


class BertLayerW(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "ln1_w", "ln1_b", "ff1_w", "ff1_b",
                                               "ff2_w", "ff2_b", "ln2_w", "ln2_b")]


class BertWeights(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int), ("n_heads", ctypes.c_int), ("hidden", ctypes.c_int),
                ("intermediate", ctypes.c_int), ("eps", ctypes.c_float), ("dtype", ctypes.c_int), ("pooling", ctypes.c_int),
                ("vocab_size", ctypes.c_int), ("max_positions", ctypes.c_int), ("type_vocab", ctypes.c_int),
                ("word_emb", ctypes.c_void_p), ("pos_emb", ctypes.c_void_p), ("type_emb", ctypes.c_void_p),
                ("emb_ln_w", ctypes.c_void_p), ("emb_ln_b", ctypes.c_void_p),
                ("layers", BertLayerW * BERT_MAX_LAYERS)]


class AtlasHipError(RuntimeError):
    pass


_lib = None
_tune = None


def lib(tuning=False):
    """The product library. tuning=True: the tuning build of the same sources (process-global knobs: scan variant, GEMM
    configuration, cycle stamps) -- never used by the product path."""
    global _lib, _tune
    if tuning:
        if _tune is None:
            _tune = _bind(TUNE_SO)
            vp, i32 = ctypes.c_void_p, ctypes.c_int
            for name, args in (("atlas_tune_set_scan_variant", [i32]), ("atlas_tune_set_gemm_cfg", [i32]), ("atlas_tune_set_gemm_diag", [i32]),
                               ("atlas_tune_set_gemm_stamps", [vp]), ("atlas_tune_set_merge_stamps", [vp]), ("atlas_tune_set_scan_stamps", [vp])):
                getattr(_tune, name).argtypes, getattr(_tune, name).restype = args, None
        return _tune
    if _lib is None:
        _lib = _bind(HIP_SO)
    return _lib


def _bind(path):
    if not os.path.exists(path):
        raise AtlasHipError(
            f"{path} is missing: the HIP extension is not built (run `python -m atlas_amd.build`). "
            "atlas_amd has no CPU fallback."
        )
    import torch  # noqa: F401  (loads torch's HIP runtime first; see module docstring)

    L = ctypes.CDLL(path)
    vp, i64, i32, f32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.atlas_abi_version.restype = i32
    L.atlas_abi_version.argtypes = []
    L.atlas_build_info.restype = ctypes.c_char_p
    L.atlas_build_info.argtypes = []
    L.atlas_scan_topk_workspace_bytes.restype = sz
    L.atlas_scan_topk_workspace_bytes.argtypes = [i64, i32, i32, i32]
    L.atlas_scan_topk.restype = i32
    L.atlas_scan_topk.argtypes = [vp, i32, vp, i64, i32, i32, i32, f32, vp, vp, vp, vp, sz, vp]
    L.atlas_scan_topk_ex.restype = i32
    L.atlas_scan_topk_ex.argtypes = [vp, i32, vp, i64, i32, i32, i32, f32, vp, vp, vp, vp, sz, vp, vp, vp]
    L.atlas_scan_topk_flags.restype = i32
    L.atlas_scan_topk_flags.argtypes = [vp, i32, vp, i64, i32, i32, i32, f32, vp, vp, vp, vp, sz, vp, vp, vp, i32]
    L.atlas_scan_topk_pack.restype = i32
    L.atlas_scan_topk_pack.argtypes = [vp, i32, vp, i64, i32, i32, i32, f32, vp, vp, vp, vp, sz, vp, vp, vp, i32, i64, i64, vp]
    L.atlas_exact_topk_workspace_bytes.restype = sz
    L.atlas_exact_topk_workspace_bytes.argtypes = [i64, i32, i32, i32]
    L.atlas_exact_topk.restype = i32
    L.atlas_exact_topk.argtypes = [vp, i32, vp, i64, i32, i32, i32, vp, vp, vp, sz, vp]
    L.atlas_pack_candidates.restype = i32
    L.atlas_pack_candidates.argtypes = [vp, vp, i64, i64, i64, vp, vp]
    L.atlas_merge_packed.restype = i32
    L.atlas_merge_packed.argtypes = [vp, i32, i32, i32, vp, vp]
    L.atlas_xchg_bytes.restype = sz
    L.atlas_xchg_bytes.argtypes = [i32, i64]
    L.atlas_xchg_create.restype = i32
    L.atlas_xchg_create.argtypes = [i32, i64, ctypes.POINTER(vp), ctypes.c_char_p]
    L.atlas_xchg_open.restype = i32
    L.atlas_xchg_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.atlas_xchg_close.restype = i32
    L.atlas_xchg_close.argtypes = [vp]
    L.atlas_xchg_destroy.restype = i32
    L.atlas_xchg_destroy.argtypes = [vp]
    L.atlas_xchg_push.restype = i32
    L.atlas_xchg_push.argtypes = [vp, i64, ctypes.POINTER(vp), i32, i32, i64, ctypes.c_uint32, vp]
    L.atlas_xchg_merge.restype = i32
    L.atlas_xchg_merge.argtypes = [vp, i32, i32, i32, i64, ctypes.c_uint32, i32, vp, vp, vp]
    L.atlas_pool_write.restype = i32
    L.atlas_pool_write.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, vp]
    L.atlas_contriever_workspace_bytes.restype = sz
    L.atlas_contriever_workspace_bytes.argtypes = [i32, i32, i32]
    L.atlas_contriever_embed.restype = i32
    L.atlas_contriever_embed.argtypes = [ctypes.POINTER(BertWeights), vp, vp, vp, i32, i32, vp, vp, sz, vp]
    L.atlas_contriever_embed_rows.restype = i32
    L.atlas_contriever_embed_rows.argtypes = [ctypes.POINTER(BertWeights), vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]
    L.atlas_slab_pmax.restype = i32
    L.atlas_slab_pmax.argtypes = [vp, i64, i32, vp, vp]
    if L.atlas_abi_version() != ABI_VERSION:
        raise AtlasHipError(f"ABI mismatch: library {L.atlas_abi_version()} vs binding {ABI_VERSION}")
    return L


def check(rc, what):
    if rc != 0:
        names = {E_BADARG: "ATLAS_E_BADARG", E_WORKSPACE: "ATLAS_E_WORKSPACE", E_UNSUPPORTED: "ATLAS_E_UNSUPPORTED"}
        raise AtlasHipError(f"{what} failed: {names.get(rc, 'hipError_t ' + str(rc))}")


def torch_dtype_code(dtype):
    import torch

    if dtype == torch.float16:
        return DT_F16
    if dtype == torch.float32:
        return DT_F32
    if dtype == torch.bfloat16:
        return DT_BF16
    return None


def scan_sources_sha256() -> str:
    """sha256 of the CODE of csrc/dscan_kernel.h + scan_kernel.h + merge_kernel.h + atlas_hip.hip (the 64-query scan AND its launch plan), comments and blank lines
    stripped: what profiles/pmc_traffic.json is keyed on -- bench.py quotes its PMC traffic only for exactly this code, and an edited comment
    does not invalidate a measurement"""
    import hashlib
    import re

    h = hashlib.sha256()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in ("dscan_kernel.h", "scan_kernel.h", "merge_kernel.h", "atlas_hip.hip"):
        text = open(os.path.join(here, f), encoding="utf-8").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                 # block comments
        text = re.sub(r"//[^\n]*", "", text)                             # line comments (no string literal of these files holds //)
        lines = [" ".join(l.split()) for l in text.split("\n")]
        h.update("\n".join(l for l in lines if l).encode())
    return h.hexdigest()
