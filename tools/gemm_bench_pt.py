"""Persistent 256-square tile (csrc/vt_gemm_pt.hip, vt_tune(8, 1)) vs gemm_pp256d_kernel (vt_tune(8, 0)) on the shapes of the path:
the fused condition K|V projection of RDT-1B (cmap 3), DINOv2-B and SigLIP-so400m qkv / fc1 (+GELU), the RDT image adaptor."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
lib = L.lib()


def timeit(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, M, N, K, dt, act=L.ACT_NONE, kv=False, r32=False):
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    bias = torch.randn(N, device=dev)
    if r32:
        resid = torch.randn(M, N, device=dev)
        cs = torch.ones(N, device=dev)
        fn = lambda: ops.gemm(a, w, bias, colscale=cs, residual=resid, out=resid, out_dtype=torch.float32)
    elif kv:
        T = (M + 63) // 64
        out = torch.empty(N // 128, T, 2, 64, 64, device=dev, dtype=dt)
        gain = torch.ones(64, device=dev)
        fn = lambda: ops.gemm(a, w, bias, out=out, out_dtype=dt, headnorm=(gain, N // 2, None, N // 2, 1e-6, 1), cmap=(3, T))
    else:
        out = torch.empty(M, N, device=dev, dtype=dt)
        fn = lambda: ops.gemm(a, w, bias, act=act, out=out, out_dtype=dt)
    n = 5 if M * N * K > 5e11 else 20
    res = []
    for on in (1, 0, 1, 0):
        lib.vt_tune(8, on)
        res.append(timeit(fn, n))
    lib.vt_tune(8, 1)
    pt, pp = min(res[0], res[2]), min(res[1], res[3])
    tf = lambda ms: 2 * M * N * K / ms / 1e9
    print(f"{name:28s} {M:6d}x{N:5d}x{K:5d}  pt {pt*1e3:8.1f} us {tf(pt):7.1f} TF/s | pp {pp*1e3:8.1f} us {tf(pp):7.1f} TF/s | {pp/pt:5.3f}x", flush=True)


bf, h = torch.bfloat16, torch.float16
case("rdt cond K|V (B=32 img)", 139968, 4096, 2048, bf, kv=True)
if "kv" in sys.argv[1:]:
    sys.exit(0)
case("rdt img adaptor fc1 gelu", 139968, 2048, 1152, bf, act=L.ACT_GELU_TANH)
case("dinov2-b qkv", 16384, 2304, 768, h)
case("dinov2-b fc1 gelu", 16384, 3072, 768, h, act=L.ACT_GELU_ERF)
case("siglip qkv", 139968, 3456, 1152, h)
case("siglip fc1 gelu", 139968, 4304, 1152, h, act=L.ACT_GELU_TANH)
case("siglip out-proj (fp32+res)", 139968, 1152, 1152, h, r32=True)
case("siglip fc2 (fp32+res)", 139968, 1152, 4352, h, r32=True)
case("dinov2-b fc2 (fp32+res)", 16384, 768, 3072, h, r32=True)
case("square 8192", 8192, 8192, 8192, bf)
