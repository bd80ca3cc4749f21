"""Yardstick only (never used by the product): what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the
path's GEMM shapes, bf16, random operands, next to vt_gemm."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops
dev = torch.device("cuda:0")
SHAPES = [(8192, 8192, 8192), (139968, 4096, 2048), (2144, 6144, 2048), (2144, 2048, 2048), (16448, 3072, 768)]
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for name, fn in (("vt_gemm", lambda: ops.gemm(a, w, out=out, out_dtype=torch.bfloat16)), ("library", lambda: torch.matmul(a, w.t(), out=out))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20 if M * N * K < 1e12 else 5
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res[name] = (ms * 1e3, 2 * M * N * K / ms / 1e9)
    print(f"M={M:6d} N={N:5d} K={K:5d}: " + "  ".join(f"{k} {v[0]:8.1f} us {v[1]:7.1f} TF/s" for k, v in res.items()), flush=True)
