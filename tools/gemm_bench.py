"""Micro-benchmark of vt_gemm on the shapes of the refinement path (bf16, random operands)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops
dev = torch.device("cuda:0")
SHAPES = [(4096, 4096, 4096), (8192, 8192, 8192), (139968, 4096, 2048), (2144, 6144, 2048), (2144, 2048, 2048), (16448, 2304, 768),
          (16448, 3072, 768), (16448, 768, 3072), (16448, 768, 768)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out=out, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20 if M * N * K < 1e12 else 5
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, out=out, out_dtype=torch.bfloat16)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"M={M:6d} N={N:5d} K={K:5d}: {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
