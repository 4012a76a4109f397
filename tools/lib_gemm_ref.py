"""What the vendor library reaches on the encoder's four GEMM shapes (a yardstick for gemm_pp_kernel, not a product path):
torch.nn.functional.linear (hipBLASLt / rocBLAS) fp16, M = 512 x 128 tokens, bias fused by the library."""
import torch
import torch.nn.functional as F

M = 65536
shapes = [("QKV", 2304, 768), ("out-proj", 768, 768), ("FFN-1", 3072, 768), ("FFN-2", 768, 3072)]
for name, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.float16) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    for _ in range(5):
        F.linear(a, w, b)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for s, e in ev:
        s.record(); F.linear(a, w, b); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    med = t[len(t) // 2]
    print(f"{name:9s} M={M} N={N} K={K}: median {med * 1e3:.1f} us  min {t[0] * 1e3:.1f} us  {2 * M * N * K / med / 1e9:.0f} TFLOP/s", flush=True)
