"""Micro-benchmark of the exact-fp32 vt_gemm (register-staged gemm_kernel) on the shapes of the controller training step (B=128, T=16):
forward / data-gradient convs as GEMMs [B*T_level, Cout] over K = k*Cin, weight gradients [Cout, k*Cin] over K = B*T_level; with the
split-K factor train.py would choose and a sweep around it.  Also torch.matmul (vendor library) in fp32 as a yardstick."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, train
dev = torch.device("cuda:0")
SHAPES = [(2048, 256, 1280), (1024, 512, 2560), (512, 512, 2560), (512, 512, 5120), (1024, 256, 5120),      # fwd / dgrad
          (256, 1280, 2048), (512, 2560, 1024), (512, 2560, 512), (512, 5120, 512), (256, 5120, 1024),      # wgrad
          (128, 10752, 512), (2048, 2048, 2048), (4096, 4096, 4096)]


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


for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * K ** -0.5
    auto = train._splits(M, N, K)
    row = []
    for s in (1, 2, 4, 8, 16):
        if K // s < 64:
            continue
        if s == 1:
            ms = t(lambda: ops.gemm(a, w))
        else:
            ms = t(lambda: train._slab_sum(ops.gemm(a, w, None, splitk=s), None, (M, N)))
        row.append(f"s{s}{'*' if s == auto else ''} {ms * 1e3:6.1f}us {2 * M * N * K / ms / 1e9:5.1f}TF")
    ms = t(lambda: torch.matmul(a, w.t()))
    print(f"M={M:5d} N={N:5d} K={K:5d}: " + " | ".join(row) + f" || torch {ms * 1e3:6.1f}us {2 * M * N * K / ms / 1e9:5.1f}TF", flush=True)
