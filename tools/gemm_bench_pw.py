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


def check_small(M, N, odt, act, res, hn):
    """vt_gemm_pws.hip (M <= 512): every k-split factor against the torch fp32 product; S = 1 without scratch as well."""
    g = torch.Generator(device=dev).manual_seed(M * 11 + N)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g)
    r = torch.randn(M, N, device=dev, generator=g).to(odt) if res else None
    hw = (1.0 + 0.1 * torch.randn(64, device=dev, generator=g)) if hn else None
    wp = ops.pack_w32(w)
    ws = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
    cnt = torch.zeros(4096, dtype=torch.int32, device=dev)
    kw = dict(act=act, residual=r, out_dtype=odt, headnorm=(hw, N, None, N, 1e-6, L.NORM_RMS_MEANSQ) if hn else None)
    ref = a.float() @ w.float().t() + b
    if hn:
        x = ref.view(M, N // 64, 64)
        ref = (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * hw).reshape(M, N)
    if act == L.ACT_GELU_TANH:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if res:
        ref = ref + r.float()
    lib.vt_tune(3, 1)
    errs = {}
    outs = {}
    for S in (1, 2, 4, 8):
        lib.vt_tune(4, S)
        o = ops.gemm(a, w, b, wp=wp, sk_ws=ws, sk_cnt=cnt, **kw).float()
        o2 = ops.gemm(a, w, b, wp=wp, sk_ws=ws, sk_cnt=cnt, **kw).float()       # second launch: the counters reset themselves
        errs[S] = (float((o - ref).abs().max()), bool(torch.equal(o, o2)))
        outs[S] = o
    lib.vt_tune(4, 0)
    noscratch = ops.gemm(a, w, b, wp=wp, **kw).float()
    # slab mode (the generic kernel's split-K contract): raw partial sums, summed here
    pre = a.float() @ w.float().t()
    slab_err = {}
    for S in (2, 8):
        sl = ops.gemm(a, w, None, wp=wp, splitk=S)
        slab_err[S] = float((sl.sum(0) - pre).abs().max())
    lib.vt_tune(3, 0)
    sl_old = ops.gemm(a, w, None, wp=wp, splitk=8)
    lib.vt_tune(3, 1)
    print(f"  small M={M} N={N} out={str(odt)[6:]} act={act} res={int(res)} hn={int(hn)}: " + " ".join(f"S{S} {e:.2e}{'' if same else ' NONDET'}" for S, (e, same) in errs.items()) +
          f" | no-scratch==S1 {bool(torch.equal(noscratch, outs[1]))} | slabs " + " ".join(f"S{S} {e:.2e}" for S, e in slab_err.items()) +
          f" (generic S8 {float((sl_old.sum(0) - pre).abs().max()):.2e}) | counters zero {int(cnt.abs().sum()) == 0}", flush=True)


def bench_small(M, N, odt):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    nw = 48
    wts = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
    wps = [ops.pack_w32(w) for w in wts]
    out = torch.empty(M, N, device=dev, dtype=odt)
    ws = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
    cnt = torch.zeros(4096, dtype=torch.int32, device=dev)
    lib.vt_tune(3, 1)
    for S in (0, 1, 2, 4, 8):
        lib.vt_tune(4, S)
        cold = graph_time(lambda i: ops.gemm(a, wts[i % nw], out=out, out_dtype=odt, wp=wps[i % nw], sk_ws=ws, sk_cnt=cnt), nw)
        print(f"  small M={M} N={N} out={str(odt)[6:]} pws S={S}: cold {cold:6.2f} us  ({N * K * 2 / cold / 1e6:5.2f} TB/s of weights)", flush=True)
    lib.vt_tune(4, 0)
    slab = torch.empty(8, M, N, device=dev, dtype=torch.float32)
    for S in (2, 4, 8):
        for on in (1, 0):
            lib.vt_tune(3, on)
            cold = graph_time(lambda i: ops.gemm(a, wts[i % nw], out=slab[:S], wp=wps[i % nw], splitk=S), nw)
            print(f"  small M={M} N={N} slab mode S={S} {'pws    ' if on else 'generic'}: cold {cold:6.2f} us  ({N * K * 2 / cold / 1e6:5.2f} TB/s of weights)", flush=True)
    lib.vt_tune(3, 1)


if __name__ == "__main__":
    print("small-M tile (vt_gemm_pws.hip)")
    for M in (67, 1, 96, 134, 201, 500):
        check_small(M, 2048, torch.bfloat16, L.ACT_NONE, False, True)
        check_small(M, 2048, torch.float32, L.ACT_NONE, True, False)
    check_small(67, 2048, torch.bfloat16, L.ACT_GELU_TANH, False, False)
    check_small(67, 6144, torch.bfloat16, L.ACT_NONE, False, True)
    bench_small(67, 2048, torch.bfloat16)
    bench_small(67, 6144, torch.bfloat16)
    bench_small(134, 2048, torch.bfloat16)
    bench_small(268, 2048, torch.bfloat16)
    if "--abl" in sys.argv:      # timing-only ablations of gemm_pw_kernel<bf16, bf16, 4> (what bounds the k-loop?)
        M, N = 2144, 2048
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        ws_ = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(NW)]
        wps = [ops.pack_w32(w) for w in ws_]
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        lib.vt_tune(2, 1); lib.vt_tune(1, 4)
        for k, name in ((0, "full kernel"), (1, "no fragment reads"), (2, "no weight loads"), (3, "no MFMAs"), (4, "no activation DMA"), (5, "no epilogue")):
            lib.vt_tune(5, k)
            t = graph_time(lambda i: ops.gemm(a, ws_[i % NW], out=out, out_dtype=torch.bfloat16, wp=wps[i % NW]), NW)
            print(f"  ablation {k} ({name:18s}): {t:6.2f} us", flush=True)
        lib.vt_tune(5, 0)
    if "--big" not in sys.argv:
        sys.exit(0)
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
