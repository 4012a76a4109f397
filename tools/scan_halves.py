"""Is a 32M-row scan slower than two 16M-row scans on this box, and if so, which half of the slab is the slow one? (product library) hipEvents around the
scan kernel: rows [0, 16M), rows [16M, 32M), all 32M, and the 8M-row quarters, alternated.
    python tools/scan_halves.py [contiguous]      # `contiguous`: the same slab copied into a hipExtMallocWithFlags(hipDeviceMallocContiguous) allocation, timed beside torch's"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from atlas_amd import _lib
from scan_policy_common import shard

L = _lib.lib()
N, B, k, D = 32_000_000, int(os.environ.get("QUERIES", "64")), 40, 768      # QUERIES=128: the GEMM-shaped pass (hipEvents bracket its launches)
slab = shard(N)
q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
ws = torch.zeros(int(L.atlas_scan_topk_workspace_bytes(N, B, D, k)), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
reps = 12
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in evs:            # (materialise the hipEvent_t handles: the C-ABI records them through the raw handle)
    a.record(); b.record()
torch.cuda.synchronize()
parts = [("rows [0, 32M)", 0, N), ("rows [0, 16M)", 0, N // 2), ("rows [16M, 32M)", N // 2, N // 2)] + [(f"rows [{i * 8}M, {i * 8 + 8}M)", i * (N // 4), N // 4) for i in range(4)]
acc = {p[0]: [] for p in parts}
for rnd in range(3):
    for name, r0, n in parts:
        ptr = slab.data_ptr() + r0 * D * 2
        for it in range(2 + reps):
            ev = evs[it - 2] if it >= 2 else (None, None)
            assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, ptr, n, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream,
                                           ev[0].cuda_event if ev[0] else None, ev[1].cuda_event if ev[1] else None, _lib.SCAN_TRUST_PMAX) == 0
        torch.cuda.synchronize()
        acc[name].append(np.mean([a.elapsed_time(b) for a, b in evs]))
if "contiguous" in sys.argv[1:]:
    import ctypes, glob
    hip = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*"))[0])
    hip.hipExtMallocWithFlags.argtypes, hip.hipExtMallocWithFlags.restype = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint], ctypes.c_int
    hip.hipMemcpy.argtypes, hip.hipMemcpy.restype = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int], ctypes.c_int
    for flag, label in ((0x4, "hipDeviceMallocContiguous"), (0x0, "hipDeviceMallocDefault (a second plain allocation)")):
        p2 = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p2), N * D * 2, flag)
        print(f"hipExtMallocWithFlags({label}) of {N * D * 2 / 1e9:.1f} GB: rc {rc}, ptr {p2.value}", flush=True)
        if rc != 0 or not p2.value:
            continue
        assert hip.hipMemcpy(p2, slab.data_ptr(), N * D * 2, 3) == 0          # hipMemcpyDeviceToDevice
        torch.cuda.synchronize()
        for rnd in range(3):
            for name, r0, n in parts[:3]:
                ptr = p2.value + r0 * D * 2
                for it in range(2 + reps):
                    ev = evs[it - 2] if it >= 2 else (None, None)
                    assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, ptr, n, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream,
                                                   ev[0].cuda_event if ev[0] else None, ev[1].cuda_event if ev[1] else None, _lib.SCAN_TRUST_PMAX) == 0
                torch.cuda.synchronize()
                acc.setdefault(label + ": " + name, []).append(np.mean([a.elapsed_time(b) for a, b in evs]))
        for name, r0, n in parts[:3]:
            t = float(np.mean(acc[label + ": " + name]))
            print(f"  {label}: {name:18s}: scan kernel {t:.4f} ms = {n * 1536 / t / 1e9 / 8:.3f} of 8 TB/s")
if "vmm" in sys.argv[1:]:
    # the same slab in ONE virtual range stitched from separately created physical chunks (hipMemAddressReserve + hipMemCreate + hipMemMap): does the
    # slow upper half follow the position inside an ALLOCATION (then chunks of a few GB each should all be of the fast kind)?
    import ctypes, glob
    hip = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*"))[0])

    class Loc(ctypes.Structure):
        _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]

    class Flags(ctypes.Structure):
        _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]

    class Prop(ctypes.Structure):
        _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", Loc), ("win32HandleMetaData", ctypes.c_void_p), ("allocFlags", Flags)]

    class Access(ctypes.Structure):
        _fields_ = [("location", Loc), ("flags", ctypes.c_int)]

    hip.hipMemAddressReserve.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_ulonglong]
    hip.hipMemCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.POINTER(Prop), ctypes.c_ulonglong]
    hip.hipMemMap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_ulonglong]
    hip.hipMemSetAccess.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Access), ctypes.c_size_t]
    hip.hipMemGetAllocationGranularity.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(Prop), ctypes.c_int]
    hip.hipMemcpy.argtypes, hip.hipMemcpy.restype = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int], ctypes.c_int
    dev = torch.cuda.current_device()
    prop = Prop(1, 0, Loc(1, dev), None, Flags(0, 0, 0))
    gran = ctypes.c_size_t()
    print("granularity rc", hip.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), 1), gran.value, flush=True)
    for chunk_gb in (4, 16):
        chunk = chunk_gb << 30
        total = -(-(N * D * 2) // chunk) * chunk
        base = ctypes.c_void_p()
        rc = hip.hipMemAddressReserve(ctypes.byref(base), total, 0, None, 0)
        print(f"chunks of {chunk_gb} GB: reserve rc {rc} base {base.value}", flush=True)
        if rc != 0:
            continue
        ok = True
        for off in range(0, total, chunk):
            h = ctypes.c_void_p()
            rc1 = hip.hipMemCreate(ctypes.byref(h), chunk, ctypes.byref(prop), 0)
            rc2 = hip.hipMemMap(base.value + off, chunk, 0, h, 0) if rc1 == 0 else -1
            ok = ok and rc1 == 0 and rc2 == 0
        acc_d = Access(Loc(1, dev), 3)
        rc3 = hip.hipMemSetAccess(base, total, ctypes.byref(acc_d), 1)
        print(f"  create / map ok {ok}, set access rc {rc3}", flush=True)
        if not ok or rc3 != 0:
            continue
        assert hip.hipMemcpy(base, slab.data_ptr(), N * D * 2, 3) == 0
        torch.cuda.synchronize()
        label = f"one virtual range of {chunk_gb} GB physical chunks"
        for rnd in range(3):
            for name, r0, n in parts[:3]:
                ptr = base.value + r0 * D * 2
                for it in range(2 + reps):
                    ev = evs[it - 2] if it >= 2 else (None, None)
                    assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, ptr, n, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream,
                                                   ev[0].cuda_event if ev[0] else None, ev[1].cuda_event if ev[1] else None, _lib.SCAN_TRUST_PMAX) == 0
                torch.cuda.synchronize()
                acc.setdefault(label + ": " + name, []).append(np.mean([a.elapsed_time(b) for a, b in evs]))
        for name, r0, n in parts[:3]:
            t = float(np.mean(acc[label + ": " + name]))
            print(f"  {label}: {name:18s}: scan kernel {t:.4f} ms = {n * 1536 / t / 1e9 / 8:.3f} of 8 TB/s", flush=True)
for name, r0, n in parts:
    t = float(np.mean(acc[name]))
    print(f"{name:18s}: scan kernel {t:.4f} ms = {n * 1536 / t / 1e9 / 8:.3f} of 8 TB/s   (rounds: {' '.join('%.4f' % x for x in acc[name])})")
print("two halves / whole: %.4f" % ((np.mean(acc['rows [0, 16M)']) + np.mean(acc['rows [16M, 32M)'])) / np.mean(acc['rows [0, 32M)'])))
