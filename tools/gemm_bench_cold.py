"""Is the per-denoise-step RDT Linear (2144 x 2048 x 2048, gemm_ppk_kernel) slower in the loop because its WEIGHTS are cold?  Time the
same launch with (a) one weight matrix (hot in L2 / Infinity Cache), (b) a rotation over 96 matrices (806 MB: every launch streams its
weights from HBM, as in the 28-layer loop), (c) the rotation with the next launch's weights touched by a small prefetch kernel on a
second stream while the current GEMM runs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops
dev = torch.device("cuda:0")
M, N, K, NW = 2144, 2048, 2048, 96
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(NW)]
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)


def run(fn, n=NW * 2):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def graph(fn, n=NW):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(4):
            fn(i)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
        for _ in range(2):
            g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


hot = lambda i: ops.gemm(a, ws[0], out=out, out_dtype=torch.bfloat16)
cold = lambda i: ops.gemm(a, ws[i % NW], out=out, out_dtype=torch.bfloat16)
print(f"graph replay, hot weights : {graph(hot):6.2f} us / launch")
print(f"graph replay, cold weights: {graph(cold):6.2f} us / launch")
side = torch.cuda.Stream()
sink = torch.zeros(1, device=dev)


def cold_pf(i):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        sink.add_(ws[(i + 1) % NW].view(torch.int16)[::1, ::64].sum())        # touches one element per 128-B line... (strided read)
    ops.gemm(a, ws[i % NW], out=out, out_dtype=torch.bfloat16)
    cur.wait_stream(side)


print(f"graph replay, cold + naive prefetch of the next weights on a side stream: {graph(cold_pf):6.2f} us / launch")
