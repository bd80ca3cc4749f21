"""What the RMSNorm hand-off costs each side (csrc/vt_gemm.h xn_out / rs_part): the residual Linear with and without the extra outputs, the consuming Linear with and
without the row scale (same operands), and the norm launch they replace — M = 2144 rows, D = 2048, graph-replayed chains of 48 launches over 24 weight sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
M, D, NW = 2144, 2048, 24
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc


def graph_time(fn, n=48):
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
        s.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for i in range(n):
                fn(i)
        gr.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            gr.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / 5 / n * 1e3


for N2 in (2048, 6144):
    a = rn(M, D).to(torch.bfloat16)
    w1 = [(rn(D, D, sc=D ** -0.5)).to(torch.bfloat16) for _ in range(NW)]
    w2 = [(rn(N2, D, sc=D ** -0.5)).to(torch.bfloat16) for _ in range(NW)]
    wp1, wp2 = [ops.pack_w32(w) for w in w1], [ops.pack_w32(w) for w in w2]
    b1, b2 = rn(D), rn(N2)
    gain = rn(D) * 0.2 + 1.0
    x = rn(M, D) * 3.0
    xo = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    part = torch.empty(M, 2 * D // 128, 2, device=dev)
    y = torch.empty(M, N2, dtype=torch.bfloat16, device=dev)
    hw = rn(64) * 0.1 + 1.0
    head = (hw, N2, None, N2, 1e-6, L.NORM_RMS_MEANSQ)
    ops.gemm(a, w1[0], b1, residual=x, out=x, out_dtype=torch.float32, wp=wp1[0], xn=(xo, gain, part))
    xs = (xo.float() * 0.05).to(torch.bfloat16)          # the same operand at the magnitude of a normalised row
    t = {}
    t["producer plain"] = graph_time(lambda i: ops.gemm(a, w1[i % NW], b1, residual=x, out=x, out_dtype=torch.float32, wp=wp1[i % NW]))
    t["producer + xn/part"] = graph_time(lambda i: ops.gemm(a, w1[i % NW], b1, residual=x, out=x, out_dtype=torch.float32, wp=wp1[i % NW], xn=(xo, gain, part)))
    t["norm launch"] = graph_time(lambda i: L.check(L.lib().vt_rownorm(L.ptr(x), L.dt_code(x.dtype), D, L.ptr(xo), L.dt_code(xo.dtype), D, L.ptr(gain), L.ptr(None), M, D, 1e-6,
                                                                     L.NORM_RMS_MEANSQ, L.stream_ptr(dev)), "rownorm"))
    t["consumer plain (A = x*gain)"] = graph_time(lambda i: ops.gemm(xo, w2[i % NW], b2, headnorm=head, wp=wp2[i % NW], out=y))
    t["consumer plain (A scaled to unit rows)"] = graph_time(lambda i: ops.gemm(xs, w2[i % NW], b2, headnorm=head, wp=wp2[i % NW], out=y))
    t["consumer + row scale"] = graph_time(lambda i: ops.gemm(xo, w2[i % NW], b2, headnorm=head, wp=wp2[i % NW], out=y, rs=(part, 1e-6)))
    print(f"N2 = {N2}:", flush=True)
    for k, v in t.items():
        print(f"   {k:42s} {v:7.2f} us", flush=True)
