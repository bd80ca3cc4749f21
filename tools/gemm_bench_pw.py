"""The per-denoise-step RDT Linears (M = 32 x 67 = 2144 rows, K = 2048, N = 2048 / 6144) on the weights-in-registers tile
(csrc/vt_gemm_pw.hip, fragment-packed W streamed global -> VGPR) against the tiles it replaces (gemm_ppk_kernel / gemm_pp256d_kernel):
correctness against a torch fp32 product of the same bf16 operands (yardstick only), then graph-replayed timing with hot weights (one
matrix) and cold weights (a rotation over 96 matrices = 806 MB, every launch streams its weights from HBM as in the 28-layer loop)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
lib = L.lib()
K = 2048
NW = 96


def check(M, N, odt, act, res, hn):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g)
    r = torch.randn(M, N, device=dev, generator=g).to(odt) if res else None
    hw = (1.0 + 0.1 * torch.randn(64, device=dev, generator=g)) if hn else None
    wp = ops.pack_w32(w)
    kw = dict(act=act, residual=r, out_dtype=odt, headnorm=(hw, N, None, N, 1e-6, L.NORM_RMS_MEANSQ) if hn else None)
    lib.vt_tune(2, 1)
    got = {}
    for nb in (4, 8):
        lib.vt_tune(1, nb)
        got[nb] = ops.gemm(a, w, b, wp=wp, **kw).float()
    lib.vt_tune(2, 0)
    old = ops.gemm(a, w, b, wp=wp, **kw).float()
    lib.vt_tune(2, 1)
    ref = a.float() @ w.float().t() + b
    if hn:
        x = ref.view(M, N // 64, 64)
        ref = (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * hw).reshape(M, N)
    if act == L.ACT_GELU_TANH:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if res:
        ref = ref + r.float()
    e = {k: float((v - ref).abs().max()) for k, v in got.items()}
    print(f"  check M={M} N={N} out={str(odt)[6:]} act={act} res={int(res)} hn={int(hn)}: |pw4-ref| {e[4]:.3e} |pw8-ref| {e[8]:.3e} |old-ref| {float((old - ref).abs().max()):.3e} "
          f"|pw4-old| {float((got[4] - old).abs().max()):.3e}  pw4==pw8 {bool(torch.equal(got[4], got[8]))}", flush=True)


def graph_time(fn, n):
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


def bench(M, N, odt, res):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    nw = NW if N == 2048 else NW // 3
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
    wps = [ops.pack_w32(w) for w in ws]
    out = torch.empty(M, N, device=dev, dtype=odt)
    r = out if res else None
    fl = 2.0 * M * N * K
    for name, knobs in (("old tile", (2, 0)), ("pw NB=4", (1, 4)), ("pw NB=8", (1, 8))):
        lib.vt_tune(2, 1)
        lib.vt_tune(*knobs)
        hot = graph_time(lambda i: ops.gemm(a, ws[0], out=out, out_dtype=odt, residual=r, wp=wps[0]), nw)
        cold = graph_time(lambda i: ops.gemm(a, ws[i % nw], out=out, out_dtype=odt, residual=r, wp=wps[i % nw]), nw)
        print(f"  M={M} N={N} out={str(odt)[6:]} res={int(res)} {name:9s}: hot {hot:6.2f} us ({fl / hot / 1e6:6.0f} TF/s)   cold {cold:6.2f} us ({fl / cold / 1e6:6.0f} TF/s)", flush=True)
    lib.vt_tune(2, 1); lib.vt_tune(1, 0)


if __name__ == "__main__":
    print("correctness (bf16 operands, fp32 torch product as yardstick)")
    for M in (2144, 2100):
        check(M, 2048, torch.bfloat16, L.ACT_NONE, False, False)
        check(M, 2048, torch.bfloat16, L.ACT_GELU_TANH, False, False)
        check(M, 2048, torch.float32, L.ACT_NONE, True, False)
        check(M, 2048, torch.bfloat16, L.ACT_NONE, False, True)
        check(M, 6144, torch.bfloat16, L.ACT_NONE, False, True)
    print("timing (hipGraph replay)")
    bench(2144, 2048, torch.bfloat16, False)
    bench(2144, 2048, torch.float32, True)
    bench(2144, 6144, torch.bfloat16, False)
