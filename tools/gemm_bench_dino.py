"""DINOv2-base Linears at the bench shape (64 images x 257 tokens = 16448 rows, fp16 storage): cost of the fused epilogues."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
M = 16448
h = torch.float16


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    a = torch.randn(M, K, device=dev).to(h)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(h)
    bias = torch.randn(N, device=dev)
    ls = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=h)
    o32 = torch.empty(M, N, device=dev)
    rows = [("plain->f16", lambda: ops.gemm(a, w, None, out=o16, out_dtype=h)),
            ("bias->f16", lambda: ops.gemm(a, w, bias, out=o16, out_dtype=h)),
            ("bias+gelu_erf->f16", lambda: ops.gemm(a, w, bias, act=L.ACT_GELU_ERF, out=o16, out_dtype=h)),
            ("bias+layerscale+residual->f32", lambda: ops.gemm(a, w, bias, colscale=ls, residual=res, out=o32, out_dtype=torch.float32)),
            ("torch.matmul f16", lambda: torch.matmul(a, w.t()))]
    for nm, fn in rows:
        ms = t(fn)
        print(f"{name:5s} {M}x{N}x{K} {nm:32s} {ms * 1e3:8.1f} us {2 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
