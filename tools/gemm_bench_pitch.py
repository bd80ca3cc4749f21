import sys, os
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops
dev = torch.device("cuda:0")
pad = int(sys.argv[1])
for (M, N, K) in [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]]:
    a = torch.randn(M, K + pad, device=dev).to(torch.bfloat16)[:, :K]
    w = (torch.randn(N, K + pad, device=dev) * K ** -0.5).to(torch.bfloat16)[:, :K]
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
    print(f"pad {pad:4d} M={M:6d} N={N:5d} K={K:5d}: {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
