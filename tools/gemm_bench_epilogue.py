"""Cost of the fused epilogue options on the cached-condition projection shape (bf16)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops
dev = torch.device("cuda:0")
B, Lc, D, H = 32, 4374, 2048, 32
M = B * Lc
T = (M + 63) // 64
a = torch.randn(M, D, device=dev).to(torch.bfloat16)
w = (torch.randn(D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
bias = torch.randn(D, device=dev)
g = torch.randn(64, device=dev)
out = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
kv = torch.empty(H, T, 2, 64, 64, device=dev, dtype=torch.bfloat16)
hn = (g, D, None, D, 1e-6, 1)
cases = {"plain": dict(out=out), "bias": dict(out=out, bias=bias), "bias+headnorm": dict(out=out, bias=bias, headnorm=hn),
         "bias+headnorm+cmap1": dict(out=kv, bias=bias, headnorm=hn, cmap=(1, T)), "bias+cmap2": dict(out=kv, bias=bias, cmap=(2, T))}
for name, kw in cases.items():
    kw = dict(kw)
    b = kw.pop("bias", None)
    f = lambda: ops.gemm(a, w, b, out_dtype=torch.bfloat16, **kw)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:24s} {ms*1e3:9.1f} us  {2*M*D*D/ms/1e9:8.1f} TF/s", flush=True)
