"""Experiment (dev tool): the refresh batch split over P streams, each confined to its own CUs (hipExtStreamCreateWithCUMask), so that the
streams' GEMM epilogues (synchronised write bursts when every CU runs the same kernel in lock-step) interleave with the other streams'
k-loops. Prints passages/s for P = 1 (plain stream), 2, 4, 8 with two mask layouts."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from atlas_amd import retrievers, _lib

L = _lib.lib()
hip = ctypes.CDLL("libamdhip64.so.7")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(cus):
    words = (NCU + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for c in cus:
        m[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


NB, LEN = 512, int(os.environ.get("LEN", "128"))
enc = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (NB, LEN), generator=g).cuda()
mask = torch.ones((NB, LEN), dtype=torch.int64).cuda()
out = torch.empty((NB, 768), dtype=torch.float16, device="cuda")
enc.embed_into(out, ids, mask)
torch.cuda.synchronize()
ref = out.clone()
w = enc._pack()


def run(P, layout, reps=6):
    if P == 1 and layout == "plain":
        streams = [torch.cuda.current_stream()]
    elif layout == "interleaved":
        streams = [masked_stream([c for c in range(NCU) if c % P == p]) for p in range(P)]
    else:
        streams = [masked_stream(range(p * NCU // P, (p + 1) * NCU // P)) for p in range(P)]
    nb = NB // P
    need = L.atlas_contriever_workspace_bytes(nb, LEN, w.dtype)
    wss = [torch.empty(int(need), dtype=torch.uint8, device="cuda") for _ in range(P)]
    o = torch.zeros_like(out)
    torch.cuda.synchronize()

    def one_pass():
        for p, s in enumerate(streams):
            a, b = p * nb, (p + 1) * nb
            rc = L.atlas_contriever_embed(ctypes.byref(w), ids[a:b].data_ptr(), mask[a:b].data_ptr(), None, nb, LEN, o[a:b].data_ptr(),
                                          wss[p].data_ptr(), wss[p].numel(), s.cuda_stream)
            assert rc == 0, rc
    one_pass(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        one_pass()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f"P={P} {layout:12s} {dt * 1e3:7.2f} ms per {NB} passages  {NB / dt:8.0f} passages/s   identical={torch.equal(o, ref)}", flush=True)


run(1, "plain")
for P in (2, 4, 8):
    for layout in ("interleaved", "blocked"):
        run(P, layout)
run(1, "plain")
