"""Does the time of the power-bound K|V projection depend on the operands' VALUES?  Same shape (139 968 x 4096 x 2048, bf16, cmap 3), operands N(0, 1) scaled by
powers of two (exact in bf16), an all-zero A (the floor: no toggling), and a constant A."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
M, N, K = 139968, 4096, 2048
T = (M + 63) // 64
bf = torch.bfloat16
a0 = torch.randn(M, K, device=dev)
w0 = torch.randn(N, K, device=dev) * K ** -0.5
bias = torch.randn(N, device=dev)
gain = torch.ones(64, device=dev)
out = torch.empty(N // 128, T, 2, 64, 64, device=dev, dtype=bf)


def t(a, w, n=6):
    fn = lambda: ops.gemm(a, w, bias, out=out, out_dtype=bf, headnorm=(gain, N // 2, None, N // 2, 1e-6, 1), cmap=(3, T))
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(2):
    for name, sa, sw in (("A x 1, W x 1", 1.0, 1.0), ("A x 2^-6", 2.0 ** -6, 1.0), ("A x 2^-3", 0.125, 1.0), ("A x 2^3", 8.0, 1.0), ("A x 2^6", 64.0, 1.0),
                         ("W x 2^5 (unit rows)", 1.0, 32.0), ("A x 2^-3, W x 2^5", 0.125, 32.0)):
        print(f"{name:24s} {t((a0 * sa).to(bf), (w0 * sw).to(bf)):8.1f} us", flush=True)
    print(f"{'A = 0':24s} {t(torch.zeros(M, K, device=dev, dtype=bf), w0.to(bf)):8.1f} us", flush=True)
    print(f"{'A = 1 (constant)':24s} {t(torch.ones(M, K, device=dev, dtype=bf), w0.to(bf)):8.1f} us", flush=True)
    print(f"{'A = |N(0,1)| (one sign)':24s} {t(a0.abs().to(bf), w0.to(bf)):8.1f} us", flush=True)
    print(f"{'A uniform(-1,1)':24s} {t((torch.rand(M, K, device=dev) * 2 - 1).to(bf), w0.to(bf)):8.1f} us", flush=True)
