"""The per-denoise-step Linear shapes of RDT in a sustained loop, with the epilogues the model uses (cold weights: 64 copies)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
M, N, K = 2144, 2048, 2048
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(64)]
bias = torch.randn(N, device=dev)
g64 = torch.randn(64, device=dev) + 1
out16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
x32 = torch.randn(M, N, device=dev)
cases = {
    "plain bf16 out": lambda w: ops.gemm(a, w, out=out16, out_dtype=torch.bfloat16),
    "bias + tanh-GELU, bf16 out (fc1)": lambda w: ops.gemm(a, w, bias, act=L.ACT_GELU_TANH, out=out16, out_dtype=torch.bfloat16),
    "bias + head RMSNorm, bf16 out (cross q)": lambda w: ops.gemm(a, w, bias, out=out16, out_dtype=torch.bfloat16, headnorm=(g64, N, None, N, 1e-6, 1)),
    "bias + fp32 residual in place (proj / fc2)": lambda w: ops.gemm(a, w, bias, residual=x32, out=x32, out_dtype=torch.float32),
}
for name, f in cases.items():
    for _ in range(3):
        f(ws[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 2000
    e0.record()
    for i in range(n):
        f(ws[i % 64])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:45s} {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s", flush=True)
