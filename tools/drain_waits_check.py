"""Digest of the outputs of the hand-scheduled GEMM tiles (gemm_pt / gemm_pp256d / gemm_pw / gemm_pws) on a set of shapes that cover every epilogue kind, ragged
edges and the smallest K — printed as one sha256 per case.  tools/drain_waits_check.sh runs it against the product library and against a debug build whose counted
`s_waitcnt vmcnt(N)` are all `vmcnt(0)`: a wait count that is too LARGE (a race the tests happen not to hit) shows up as a differing digest."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
L.lib()


def rnd(shape, seed, dt=torch.float32, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dt)


def digest(name, t):
    torch.cuda.synchronize()
    print(name, hashlib.sha256(t.contiguous().cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:20], flush=True)


for dt in (torch.bfloat16, torch.float16):
    tag = "bf16" if dt == torch.bfloat16 else "f16"
    for (M, N, K, act) in [(16384, 2304, 768, L.ACT_NONE), (16500, 3072, 768, L.ACT_GELU_ERF), (35000, 2048, 576, L.ACT_GELU_TANH), (65537, 2112, 512, L.ACT_GELU_ERF),
                           (33023, 1088, 512, L.ACT_NONE), (2144, 6144, 2048, L.ACT_NONE)]:
        a, w, b = rnd((M, K), 1, dt), rnd((N, K), 2, dt, K ** -0.5), rnd((N,), 3)
        for rep in range(2):
            digest(f"p16 {tag} {M}x{N}x{K} act{act} #{rep}", ops.gemm(a, w, b, act=act, out_dtype=dt))
    for (M, N, K) in [(69984, 1152, 1152), (16448, 768, 3072), (33023, 2112, 512)]:
        a, w, b, cs, r = rnd((M, K), 1, dt), rnd((N, K), 2, dt, K ** -0.5), rnd((N,), 3), rnd((N,), 4) * 0.3 + 1, rnd((M, N), 5)
        for rep in range(2):
            digest(f"r32 {tag} {M}x{N}x{K} #{rep}", ops.gemm(a, w, b, colscale=cs, residual=r, out_dtype=torch.float32))
    for (M, K, mode) in [(70001, 512, L.NORM_RMS_MEANSQ), (34992, 2048, L.NORM_RMS_VAR), (33023, 512, L.NORM_RMS_MEANSQ)]:
        N, T = 4096, (M + 63) // 64
        a, w, b, g = rnd((M, K), 1, dt), rnd((N, K), 2, dt, K ** -0.5), rnd((N,), 3), rnd((64,), 4) * 0.2 + 1
        out = torch.zeros(N // 128, T, 2, 64, 64, dtype=dt, device=dev)
        for rep in range(2):
            out.zero_()
            ops.gemm(a, w, b, out=out, out_dtype=dt, headnorm=(g, N // 2, None, N // 2, 1e-6, mode), cmap=(3, T))
            digest(f"kv {tag} {M}x{K} mode{mode} #{rep}", out)
    # weights-in-registers tile (M = 2144: one round) with the RMSNorm hand-off, and the small-M packed tile
    D = 2048
    for (M, N2) in [(2144, 6144), (2100, 2048)]:
        a, w1, b1, x0, gain = rnd((M, D), 1, dt), rnd((D, D), 2, dt, D ** -0.5), rnd((D,), 3), rnd((M, D), 4) * 3 + 2, rnd((D,), 5) * 0.2 + 1
        w2, b2 = rnd((N2, D), 6, dt, D ** -0.5), rnd((N2,), 7)
        wp1, wp2 = ops.pack_w32(w1), ops.pack_w32(w2)
        for rep in range(2):
            x = x0.clone()
            xo = torch.zeros(M, D, dtype=dt, device=dev)
            part = torch.zeros(M, 2 * D // 128, 2, device=dev)
            ops.gemm(a, w1, b1, residual=x, out=x, out_dtype=torch.float32, wp=wp1, xn=(xo, gain, part))
            digest(f"pw-producer {tag} {M} #{rep}", x); digest(f"pw-xn {tag} {M} #{rep}", xo)
            digest(f"pw-consumer {tag} {M}x{N2} #{rep}", ops.gemm(xo, w2, b2, wp=wp2, rs=(part, 1e-6, L.NORM_RMS_VAR)))
            digest(f"pw-plain {tag} {M}x{N2} #{rep}", ops.gemm(a, w2, b2, act=L.ACT_GELU_TANH, wp=wp2))
    for M in (67, 134, 400):
        a, w, b = rnd((M, D), 1, dt), rnd((D, D), 2, dt, D ** -0.5), rnd((D,), 3)
        wp = ops.pack_w32(w)
        for rep in range(2):
            digest(f"pws {tag} {M} #{rep}", ops.gemm(a, w, b, wp=wp, out_dtype=torch.float32))
