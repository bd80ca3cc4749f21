import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/vla-touch_amd"]
import torch
from vlatouch import ops
dev = torch.device("cuda:0")
M, N, K = 2144, 2048, 2048
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(64)]   # 64 different weights (537 MB): cold-ish
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for n, cold in ((20, False), (4000, False), (4000, True)):
    for _ in range(3): ops.gemm(a, ws[0], out=out, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.gemm(a, ws[i % 64] if cold else ws[0], out=out, out_dtype=torch.bfloat16)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n} cold={cold}: {e0.elapsed_time(e1)/n*1e3:.1f} us")
