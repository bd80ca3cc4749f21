"""GPU parity: primitive kernels (through the C ABI) against plain torch fp32 references.
Tolerances: fp32 path 1e-4-class (relative to output scale), bf16 path 1e-2-class."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import cases

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from vlatouch import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rnd(shape, seed, dev, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def test_mfma_fragment_layouts(dev):
    from vlatouch import ops
    e_bf16, e_f32 = ops.mfma_selftest(dev)
    assert e_bf16 == 0.0 and e_f32 == 0.0, (e_bf16, e_f32)


ACTS = {0: lambda x: x, 1: lambda x: F.gelu(x), 2: lambda x: F.gelu(x, approximate="tanh"), 3: F.silu, 4: F.mish}


# (2144, 2048, 1088): one round of 160x128 tiles -> the in-block split-K ping-pong kernel, odd number of k-tiles (17)
@pytest.mark.parametrize("M,N,K", [(32, 256, 784), (70, 100, 40), (512, 512, 1280), (1000, 768, 592), (4112, 2304, 768), (33, 10, 256), (2144, 2048, 1088),
                                   (2100, 1920, 1024)])
@pytest.mark.parametrize("mode", ["f32", "bf16", "a32w16", "f16"])
def test_gemm_plain(dev, M, N, K, mode):
    from vlatouch import ops
    lo = torch.float16 if mode == "f16" else torch.bfloat16
    adt = lo if mode in ("bf16", "f16") else torch.float32
    wdt = torch.float32 if mode == "f32" else lo
    a = rnd((M, K), 1, dev, adt)
    w = rnd((N, K), 2, dev, wdt, K ** -0.5)
    bias = rnd((N,), 3, dev)
    cs = rnd((N,), 4, dev) + 1.0
    act = (M + N) % 5
    for odt in ((torch.float32,) if mode == "f32" else (torch.float32, lo)):
        res = rnd((M, N), 5, dev, odt)
        ref = res.float() + cs * ACTS[act](a.float() @ w.float().t() + bias)
        if mode == "a32w16":
            ref = res.float() + cs * ACTS[act](a.to(torch.bfloat16).float() @ w.float().t() + bias)
        out = ops.gemm(a, w, bias, act=act, colscale=cs, residual=res, out_dtype=odt)
        tol = 2e-5 if (mode == "f32") else (2e-3 if odt == torch.float32 else (2e-3 if mode == "f16" else 1e-2))
        assert rel_err(out.float(), ref) < tol, (mode, odt, rel_err(out.float(), ref))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_gemm_splitk_slabs(dev, mode):
    from vlatouch import ops
    dt = torch.float32 if mode == "f32" else torch.bfloat16
    a, w = rnd((256, 2560), 1, dev, dt), rnd((512, 2560), 2, dev, dt, 2560 ** -0.5)
    slabs = ops.gemm(a, w, splitk=5)
    ref = a.float() @ w.float().t()
    assert rel_err(slabs.sum(0), ref) < (2e-5 if mode == "f32" else 2e-3)


def _pack_conv(w, cin_pad):
    from vlatouch.engine import _conv_tapmajor
    return _conv_tapmajor(w.cpu().float(), cin_pad)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [dict(cin=16, cout=256, k=5, T=16, stride=1), dict(cin=256, cout=512, k=5, T=8, stride=1),
                                 dict(cin=512, cout=512, k=3, T=8, stride=2), dict(cin=1024, cout=256, k=5, T=12, stride=1)])
def test_conv1d_implicit_gemm(dev, mode, cfg):
    from vlatouch import ops
    dt = torch.float32 if mode == "f32" else torch.bfloat16
    B, cin, cout, k, T, stride = 3, cfg["cin"], cfg["cout"], cfg["k"], cfg["T"], cfg["stride"]
    x = rnd((B, T, cin), 1, dev, dt)
    w = rnd((cout, cin, k), 2, dev, torch.float32, (cin * k) ** -0.5)
    b = rnd((cout,), 3, dev)
    wp = _pack_conv(w, cin).to(dt).to(dev)
    pad = k // 2
    tout = (T + 2 * pad - k) // stride + 1
    out = ops.conv1d_cl(x, wp, b, taps=k, cin=cin, tout=tout, stride=stride, off0=-pad, out_dtype=torch.float32)
    ref = F.conv1d(x.float().transpose(1, 2), w.to(dt).float(), b, stride=stride, padding=pad).transpose(1, 2)
    assert rel_err(out, ref) < (2e-5 if mode == "f32" else 3e-3)


def test_conv_transpose_parity_split(dev):
    from vlatouch import ops
    from vlatouch.engine import _convT_parity
    B, Cc, T = 2, 256, 8
    x = rnd((B, T, Cc), 1, dev)
    w = rnd((Cc, Cc, 4), 2, dev, torch.float32, (Cc * 2) ** -0.5)
    b = rnd((Cc,), 3, dev)
    ref = F.conv_transpose1d(x.transpose(1, 2), w, b, stride=2, padding=1).transpose(1, 2)      # [B, 2T, C]
    out = torch.empty(B, 2 * T, Cc, device=dev)
    even = ops.conv1d_cl(x, _convT_parity(w.cpu(), (1, 3)).to(dev), b, taps=2, cin=Cc, tout=T, off0=0, tstep=-1)
    odd = ops.conv1d_cl(x, _convT_parity(w.cpu(), (0, 2)).to(dev), b, taps=2, cin=Cc, tout=T, off0=1, tstep=-1)
    out[:, 0::2], out[:, 1::2] = even, odd
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("Nq,Nk,masked", [(257, 257, False), (67, 67, False), (67, 40, True), (67, 1000, False), (5, 730, False)])
def test_attention(dev, mode, Nq, Nk, masked):
    from vlatouch import ops
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[mode]
    B, H = 2, 3
    qkv = rnd((B, max(Nq, Nk), 3, H, 64), 1, dev, dt)
    q, k, v = qkv[:, :Nq, 0], qkv[:, :Nk, 1], qkv[:, :Nk, 2]
    mask = None
    if masked:
        mask = torch.ones(B, Nk, dtype=torch.bool, device=dev)
        mask[0, Nk - 7:] = False
        mask[1, 3:9] = False
    out = ops.attention(q, k, v, kmask=mask)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) / 8.0
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float()).reshape(B, Nq, H * 64)
    assert rel_err(out.float(), ref) < {"f32": 3e-5, "bf16": 1.5e-2, "f16": 2e-3}[mode]


@pytest.mark.parametrize("T,Cc", [(16, 256), (8, 512), (4, 512), (48, 256), (64, 256)])
def test_groupnorm_mish_film_residual(dev, T, Cc):
    from vlatouch import ops
    B, S = 3, 3
    slabs = rnd((S, B * T, Cc), 1, dev)
    bias, gamma, beta = rnd((Cc,), 2, dev), rnd((Cc,), 3, dev) + 1, rnd((Cc,), 4, dev)
    film = rnd((B, 2 * Cc), 5, dev)
    res = rnd((B * T, Cc), 6, dev)
    y = (slabs.sum(0) + bias).reshape(B, T, Cc).transpose(1, 2)
    y = F.mish(F.group_norm(y, 8, gamma, beta, 1e-5))
    ref1 = (film[:, :Cc, None] * y + film[:, Cc:, None]).transpose(1, 2).reshape(B * T, Cc)
    ref2 = y.transpose(1, 2).reshape(B * T, Cc) + res
    assert rel_err(ops.groupnorm_cl(slabs, bias, gamma, beta, B=B, T=T, film=film), ref1) < 2e-5
    assert rel_err(ops.groupnorm_cl(slabs, bias, gamma, beta, B=B, T=T, residual=res), ref2) < 2e-5


# rows >= 8192: the wave-per-row kernel of the ViT towers (SigLIP 1152-wide, DINOv2 768-wide rows)
@pytest.mark.parametrize("D,rows", [(256, 37), (384, 37), (768, 37), (2048, 37), (2048, 300), (1152, 261), (1152, 9001), (768, 8200), (2048, 8193)])
def test_rownorm_modes(dev, D, rows):
    from vlatouch import ops, _lib as L
    x, w, b = rnd((rows, D), 1, dev), rnd((D,), 2, dev) + 1, rnd((D,), 3, dev)
    assert rel_err(ops.rownorm(x, w, b, 1e-6), F.layer_norm(x, (D,), w, b, 1e-6)) < 2e-5
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    assert rel_err(ops.rownorm(x, w, None, 1e-6, L.NORM_RMS_MEANSQ), ref) < 2e-5
    ref = x * torch.rsqrt(x.var(-1, keepdim=True) + 1e-6) * w
    assert rel_err(ops.rownorm(x, w, None, 1e-6, L.NORM_RMS_VAR), ref) < 2e-5
    yb = ops.rownorm(x, w, b, 1e-6, out_dtype=torch.bfloat16)
    assert rel_err(yb.float(), F.layer_norm(x, (D,), w, b, 1e-6)) < 1e-2


def test_headnorm_inplace(dev):
    from vlatouch import ops
    x = rnd((2, 67, 3, 4, 64), 1, dev)            # fused qkv buffer [B, N, 3, H, 64]
    w = rnd((64,), 2, dev) + 1
    ref = x.clone()
    ref[:, :, 0] = x[:, :, 0] * torch.rsqrt(x[:, :, 0].pow(2).mean(-1, keepdim=True) + 1e-6) * w
    ops.headnorm_(x, 4, w, tok_stride=3 * 4 * 64, tokens=2 * 67)      # q slice: offset 0
    assert rel_err(x, ref) < 2e-5


def test_action_normalize_matches_golden(dev):
    from vlatouch.engine import action_normalize
    g = np.load(cases.GOLDEN + "/g7_norm.npz")
    st = cases.stats("nontrivial")
    a = cases.predict_inputs(2, 16, 224)["vla"].to(dev)
    for kind, key in (("vla", "n_vla"), ("expert", "n_exp")):
        mn, mx = (st["vla_mins"], st["vla_maxs"]) if kind == "vla" else (st["action_mins"], st["action_maxs"])
        out = action_normalize(a, mn, mx, denorm=False)
        assert np.abs(out.cpu().numpy() - g[key]).max() < 2e-6
        out = action_normalize(a, mn, mx, denorm=True)
        assert np.abs(out.cpu().numpy() - g["d_" + key[2:]]).max() < 2e-6


@pytest.mark.parametrize("M,N,K", [(2144, 6144, 2048), (4374, 4096, 2048), (1000, 768, 3072), (2144, 2048, 2048), (35000, 2048, 512)])
def test_gemm_large_path_with_fused_headnorm(dev, M, N, K):
    """The LDS-DMA large-GEMM kernel incl. its fused per-head RMSNorm epilogue (q_norm | k_norm | v untouched)."""
    from vlatouch import ops, _lib as L
    a = rnd((M, K), 1, dev, torch.bfloat16)
    w = rnd((N, K), 2, dev, torch.bfloat16, K ** -0.5)
    bias = rnd((N,), 3, dev)
    w0, w1 = rnd((64,), 4, dev) + 1, rnd((64,), 5, dev) + 1
    y = a.float() @ w.float().t() + bias
    plain = ops.gemm(a, w, bias, out_dtype=torch.float32)
    assert rel_err(plain, y) < 2e-3
    if ((M + 127) // 128) * ((N + 127) // 128) < 96:      # too few tiles for the large-GEMM path: no fused head-norm there
        with pytest.raises(L.VtError):
            ops.gemm(a, w, bias, out_dtype=torch.float32, headnorm=(w0, 64, None, 64, 1e-6, L.NORM_RMS_MEANSQ))
        return
    c0, c1 = (N // 3 // 64) * 64, 2 * (N // 3 // 64) * 64
    for mode in (L.NORM_RMS_MEANSQ, L.NORM_RMS_VAR):
        ref = y.clone().reshape(M, N // 64, 64)
        h0, h1 = c0 // 64, c1 // 64
        var = (lambda t: t.pow(2).mean(-1, keepdim=True)) if mode == L.NORM_RMS_MEANSQ else (lambda t: t.var(-1, keepdim=True))
        ref[:, :h0] = ref[:, :h0] * torch.rsqrt(var(ref[:, :h0]) + 1e-6) * w0
        ref[:, h0:h1] = ref[:, h0:h1] * torch.rsqrt(var(ref[:, h0:h1]) + 1e-6) * w1
        out = ops.gemm(a, w, bias, out_dtype=torch.float32, headnorm=(w0, c0, w1, c1, 1e-6, mode))
        assert rel_err(out, ref.reshape(M, N)) < 3e-3, mode
        outb = ops.gemm(a, w, bias, out_dtype=torch.bfloat16, headnorm=(w0, c0, w1, c1, 1e-6, mode))
        assert rel_err(outb.float(), ref.reshape(M, N)) < 1e-2


@pytest.mark.parametrize("M,K", [(900, 256), (70001, 512)])          # 128-column kernel / 256-square ping-pong kernel
def test_gemm_condition_tile_stream_outputs(dev, M, K):
    """cmap 1 / 2: the K / Vt halves of the cached-condition tile stream written from the GEMM epilogue == the row-major GEMM
    re-laid out on the host (tile(h, t) = [K: 64 rows x 64 d][Vt: 64 d x 64 rows in MFMA k order])."""
    from vlatouch import ops
    N, H = 2048, 32
    T = (M + 63) // 64
    a = rnd((M, K), 1, dev, torch.bfloat16)
    w = rnd((N, K), 2, dev, torch.bfloat16, K ** -0.5)
    bias = rnd((N,), 3, dev)
    ref = ops.gemm(a, w, bias, out_dtype=torch.bfloat16)                       # [M, N] bf16, same kernels, plain layout
    pad = torch.zeros(T * 64, N, dtype=torch.bfloat16, device=dev)
    pad[:M] = ref
    rows = pad.reshape(T, 64, H, 64)                                           # [t, r, h, d]
    kk = torch.arange(64)
    pos = (kk & 32) | (((kk >> 2) & 3) << 3) | (((kk >> 4) & 1) << 2) | (kk & 3)   # vt_kpos
    kv = torch.full((H, T, 2, 64, 64), 7.0, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, bias, out=kv, out_dtype=torch.bfloat16, cmap=(1, T))
    ops.gemm(a, w, bias, out=kv, out_dtype=torch.bfloat16, cmap=(2, T))
    valid = (torch.arange(T * 64, device=dev) < M).reshape(T, 64)
    got_k = kv[:, :, 0].permute(1, 2, 0, 3)                                    # [t, r, h, d]
    assert torch.equal(got_k[valid], rows[valid])
    got_v = torch.empty(T, 64, H, 64, dtype=torch.bfloat16, device=dev)        # undo the k order: row kk sits at pos[kk]
    got_v[:, kk] = kv[:, :, 1][:, :, :, pos.to(dev)].permute(1, 3, 0, 2)       # [h,t,d,kk] -> [t,kk,h,d]
    assert torch.equal(got_v[valid], rows[valid])


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("Nq,Nk", [(729, 729), (36, 36), (100, 257)])
def test_attention_head_dim_96(dev, mode, Nq, Nk):
    """hd = 96 (SigLIP's 72-wide heads zero-padded): scale passed explicitly as 72 ** -0.5."""
    from vlatouch import ops
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[mode]
    B, H = 2, 3
    qkv = rnd((B, max(Nq, Nk), 3, H, 96), 1, dev, dt)
    qkv[..., 72:] = 0
    q, k, v = qkv[:, :Nq, 0], qkv[:, :Nk, 1], qkv[:, :Nk, 2]
    out = ops.attention(q, k, v, scale=72 ** -0.5)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * 72 ** -0.5
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float()).reshape(B, Nq, H * 96)
    assert rel_err(out.float(), ref) < {"f32": 3e-5, "bf16": 1.5e-2, "f16": 2e-3}[mode]
    assert float(out.float().reshape(B, Nq, H, 96)[..., 72:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("Nq,Nk", [(729, 729), (36, 36), (100, 257), (67, 130)])
def test_attention_head_dim_80(dev, mode, Nq, Nk):
    """hd = 80 (SigLIP's 72-wide heads zero-padded to the next multiple of 16 in the 16-bit modes): two 32-deep MFMA steps + one 16-deep step
    for q.k, five 16-row output tiles; the LAST head's rows end exactly at the tensor's end (the DMA must not read past them)."""
    from vlatouch import ops
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[mode]
    B, H = 2, 3
    qkv = rnd((B, max(Nq, Nk), 3, H, 80), 7, dev, dt)
    qkv[..., 72:] = 0
    q, k, v = qkv[:, :Nq, 0], qkv[:, :Nk, 1], qkv[:, :Nk, 2]
    out = ops.attention(q, k, v, scale=72 ** -0.5)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * 72 ** -0.5
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float()).reshape(B, Nq, H * 80)
    assert rel_err(out.float(), ref) < {"bf16": 1.5e-2, "f16": 2e-3}[mode]
    assert float(out.float().reshape(B, Nq, H, 80)[..., 72:].abs().max()) == 0.0
    with pytest.raises(Exception):
        ops.attention(q.float(), k.float(), v.float(), scale=72 ** -0.5)          # fp32 keeps 96-wide padding


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("N,hd,B,H", [(729, 80, 3, 4), (257, 64, 5, 3), (730, 64, 2, 3), (1370, 64, 1, 6), (128, 64, 2, 2), (200, 80, 2, 2)])
def test_grouped_query_attention_is_bit_identical_to_one_group_per_wave(dev, mode, N, hd, B, H):
    """attn16g_kernel (a block walks the key tiles ONCE for G groups of 16 query rows per wave: one block per (image, head) for SigLIP's 729 and
    DINOv2's 257 tokens) against attn16u_kernel (one group per wave, six / three blocks per (image, head)): per query row the two kernels do the same
    operations in the same order, so the outputs must be BIT-identical — for G chosen by the launcher and for G pinned to 3 and 6 (partly filled
    last blocks, groups of padding rows skipped, a ragged last key tile) — and equal to a torch fp32 reference within the 16-bit tolerance."""
    from vlatouch import ops, _lib as L
    lib = L.lib()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[mode]
    real = 72 if hd == 80 else hd
    qkv = rnd((B, N, 3, H, hd), 11, dev, dt)
    if hd == 80:
        qkv[..., 72:] = 0
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    outs = {}
    try:
        for knob in (0, 1, 3, 6):
            L.check(lib.vt_tune(9, knob), "tune")
            outs[knob] = ops.attention(q, k, v, scale=real ** -0.5)
            torch.cuda.synchronize()
    finally:
        lib.vt_tune(9, 1)
    for knob in (1, 3, 6):
        assert torch.equal(outs[knob], outs[0]), (knob, float((outs[knob].float() - outs[0].float()).abs().max()))
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * real ** -0.5
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float()).reshape(B, N, H * hd)
    assert rel_err(outs[1].float(), ref) < {"bf16": 1.5e-2, "f16": 2e-3}[mode]


@pytest.mark.parametrize("M,N,K,splitk", [(70, 100, 64, 1), (33, 64, 96, 1), (200, 260, 1056, 1), (512, 512, 2560, 3), (64, 1280, 2048, 8), (257, 36, 160, 2)])
def test_fp32_ring_gemm_edges(dev, M, N, K, splitk):
    """vt_gemm_f32r.hip (bias-only exact-fp32 products with few blocks per CU): ragged M / N (clamped rows, scalar stores when N % 4 != 0),
    fewer k-tiles than ring slots, uneven split-K slices."""
    from vlatouch import ops
    a, w, bias = rnd((M, K), 1, dev), rnd((N, K), 2, dev, torch.float32, K ** -0.5), rnd((N,), 3, dev)
    ref = a.double().cpu() @ w.double().cpu().t()
    if splitk == 1:
        out = ops.gemm(a, w, bias)
        assert rel_err(out, ref + bias.double().cpu()) < 2e-6
        assert rel_err(ops.gemm(a, w), ref) < 2e-6
    else:
        slabs = ops.gemm(a, w, splitk=splitk)
        assert slabs.shape == (splitk, M, N) and rel_err(slabs.sum(0), ref) < 2e-6


@pytest.mark.parametrize("cfg", [dict(cin=64, cout=96, k=5, T=16, stride=1, splitk=1), dict(cin=64, cout=96, k=3, T=16, stride=2, splitk=1),
                                 dict(cin=512, cout=256, k=5, T=4, stride=1, splitk=4), dict(cin=32, cout=40, k=5, T=8, stride=1, splitk=1)])
def test_fp32_ring_conv_edges(dev, cfg):
    """The ring kernel's conv mode: taps that fall in the padding read the zero page; strided outputs; split-K over (tap, channel) tiles."""
    from vlatouch import ops
    B, cin, cout, k, T, stride, sk = 9, cfg["cin"], cfg["cout"], cfg["k"], cfg["T"], cfg["stride"], cfg["splitk"]
    x = rnd((B, T, cin), 1, dev)
    w = rnd((cout, cin, k), 2, dev, torch.float32, (cin * k) ** -0.5)
    b = rnd((cout,), 3, dev)
    wp = _pack_conv(w, cin).to(dev)
    pad = k // 2
    tout = (T + 2 * pad - k) // stride + 1
    ref = F.conv1d(x.double().cpu().transpose(1, 2), w.double().cpu(), None, stride=stride, padding=pad).transpose(1, 2)
    if sk == 1:
        out = ops.conv1d_cl(x, wp, b, taps=k, cin=cin, tout=tout, stride=stride, off0=-pad)
        assert rel_err(out, ref + b.double().cpu()) < 2e-6
    else:
        slabs = ops.conv1d_cl(x, wp, None, taps=k, cin=cin, tout=tout, stride=stride, off0=-pad, splitk=sk)
        assert rel_err(slabs.sum(0), ref) < 2e-6


# ---------------------------------------------------------------------------------------- round 3: frozen, fragment-packed weights
def _ref_linear(a, w, bias, hw, act, res):
    ref = a.float() @ w.float().t() + bias
    if hw is not None:
        M, N = ref.shape
        x = ref.view(M, N // 64, 64)
        ref = (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * hw).reshape(M, N)
    ref = ACTS[act](ref)
    return ref if res is None else ref + res.float()


@pytest.mark.parametrize("M,N", [(2144, 2048), (2100, 2048), (2144, 6144), (3000, 2048)])
@pytest.mark.parametrize("variant", ["bf16", "bf16_gelu", "bf16_headnorm", "f32_residual"])
def test_gemm_packed_weights_tile(dev, M, N, variant):
    """vt_gemm_pw.hip (weights streamed global -> VGPR from the vt_pack_w32 copy): both ring depths against the torch product and against each
    other (bit-equal: the ring depth changes the schedule, not the arithmetic); M % 160 != 0 exercises the clamped last row tile;
    (3000, 2048) = 19 x 16 tiles is not a whole round -> the dispatcher must keep the old tiles (same result either way)."""
    from vlatouch import ops, _lib as L
    K = 2048
    a = rnd((M, K), 11, dev, torch.bfloat16)
    w = rnd((N, K), 12, dev, torch.bfloat16, K ** -0.5)
    bias = rnd((N,), 13, dev)
    wp = ops.pack_w32(w)
    hw = (1.0 + 0.1 * rnd((64,), 14, dev)) if variant == "bf16_headnorm" else None
    odt = torch.float32 if variant == "f32_residual" else torch.bfloat16
    res = rnd((M, N), 15, dev, odt) if variant == "f32_residual" else None
    act = 2 if variant == "bf16_gelu" else 0
    kw = dict(act=act, residual=res, out_dtype=odt, headnorm=(hw, N, None, N, 1e-6, L.NORM_RMS_MEANSQ) if hw is not None else None)
    lib = L.lib()
    try:
        outs = []
        for nb in (4, 8):
            lib.vt_tune(1, nb)
            outs.append(ops.gemm(a, w, bias, wp=wp, **kw))
        lib.vt_tune(2, 0)
        old = ops.gemm(a, w, bias, wp=wp, **kw)
    finally:
        lib.vt_tune(1, 0); lib.vt_tune(2, 1)
    ref = _ref_linear(a, w, bias, hw, act, res)
    tol = 2e-4 if odt == torch.float32 else 1e-2
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], ref) < tol, rel_err(outs[0], ref)
    assert rel_err(old, ref) < tol


def test_pack_w32_layout(dev):
    """vt_pack_w32: out[((n/32 * K/16 + k/16) * 64 + (k%16)/8 * 32 + n%32) * 8 + k%8] = W[n][k]."""
    from vlatouch import ops
    N, K = 96, 80
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K).to(torch.bfloat16).to(dev)     # values < 2^13 are exact in bf16? no: use indices mod 256
    w = (torch.arange(N * K).reshape(N, K) % 251).to(torch.bfloat16).to(dev)
    got = ops.pack_w32(w).cpu().float().reshape(N // 32, K // 16, 2, 32, 8)
    want = w.cpu().float().reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4)
    assert torch.equal(got, want)


@pytest.mark.parametrize("M", [1, 67, 96, 134, 500])
@pytest.mark.parametrize("variant", ["bf16_headnorm", "f32_residual", "bf16_gelu"])
def test_gemm_small_m_packed_weights_tile(dev, M, variant):
    """vt_gemm_pws.hip: direct mode (S = 1, full epilogue), the in-launch ticket reduction (S = 2, 4, 8; deterministic: a second launch is
    bit-equal, the counters reset themselves) and the slab mode behind the generic split-K contract."""
    from vlatouch import ops, _lib as L
    N, K = 2048, 2048
    a = rnd((M, K), 21, dev, torch.bfloat16)
    w = rnd((N, K), 22, dev, torch.bfloat16, K ** -0.5)
    bias = rnd((N,), 23, dev)
    wp = ops.pack_w32(w)
    hw = (1.0 + 0.1 * rnd((64,), 24, dev)) if variant == "bf16_headnorm" else None
    odt = torch.float32 if variant == "f32_residual" else torch.bfloat16
    res = rnd((M, N), 25, dev, odt) if variant == "f32_residual" else None
    act = 2 if variant == "bf16_gelu" else 0
    kw = dict(act=act, residual=res, out_dtype=odt, headnorm=(hw, N, None, N, 1e-6, L.NORM_RMS_MEANSQ) if hw is not None else None)
    ref = _ref_linear(a, w, bias, hw, act, res)
    tol = 2e-4 if odt == torch.float32 else 1e-2
    ws = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
    cnt = torch.zeros(4096, dtype=torch.int32, device=dev)
    lib = L.lib()
    try:
        for S in (1, 2, 4, 8):
            lib.vt_tune(4, S)
            o1 = ops.gemm(a, w, bias, wp=wp, sk_ws=ws, sk_cnt=cnt, **kw)
            o2 = ops.gemm(a, w, bias, wp=wp, sk_ws=ws, sk_cnt=cnt, **kw)
            assert torch.equal(o1, o2), S
            assert rel_err(o1, ref) < tol, (S, rel_err(o1, ref))
        assert int(cnt.abs().sum()) == 0
        lib.vt_tune(4, 0)
        pre = a.float() @ w.float().t()
        for S in (2, 8):
            slabs = ops.gemm(a, w, None, wp=wp, splitk=S)
            assert slabs.shape == (S, M, N) and rel_err(slabs.sum(0), pre) < 2e-5
    finally:
        lib.vt_tune(4, 0)


def test_gemm_rowsplit_keeps_fused_headnorm(dev):
    """A head-norm-fused GEMM whose M lands in the exact-row-split window of the 256-square tile (M % 256 in 1..64 at a round-crossing tile
    count: 8193 rows x 6144 columns = 33 x 24 tiles) must not be split (the <= 64-row remainder has no fused-head-norm tile)."""
    from vlatouch import ops, _lib as L
    M, N, K = 8193, 6144, 512
    a = rnd((M, K), 31, dev, torch.bfloat16)
    w = rnd((N, K), 32, dev, torch.bfloat16, K ** -0.5)
    bias = rnd((N,), 33, dev)
    hw = 1.0 + 0.1 * rnd((64,), 34, dev)
    out = ops.gemm(a, w, bias, headnorm=(hw, N, None, N, 1e-6, L.NORM_RMS_MEANSQ))
    assert rel_err(out, _ref_linear(a, w, bias, hw, 0, None)) < 1e-2


def test_device_rng_slice_cast_and_cast(dev):
    from vlatouch import ops
    rng = ops.DeviceRng(1234, dev)
    x = torch.empty(1 << 20, dtype=torch.float32, device=dev)
    rng.normal_(x)
    y = torch.empty_like(x)
    rng.normal_(y)                                                  # the counter advanced on the device: a different draw
    assert not torch.equal(x, y)
    for v in (x, y):
        assert abs(float(v.mean())) < 5e-3 and abs(float(v.std()) - 1.0) < 5e-3
        assert abs(float((v ** 4).mean()) - 3.0) < 0.05            # kurtosis of a Gaussian
    assert abs(float((x * y).mean())) < 5e-3
    again = ops.DeviceRng(1234, dev).normal_(torch.empty_like(x))
    assert torch.equal(again, x)                                    # counter-based: same key + counter -> same values
    z = ops.DeviceRng(7, dev).normal_(torch.empty(1001, dtype=torch.float32, device=dev), round_bf16=True)
    assert torch.equal(z, z.to(torch.bfloat16).float())
    # replayed from a captured graph the draw advances without host involvement
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    buf = torch.empty(4096, dtype=torch.float32, device=dev)
    with torch.cuda.stream(s):
        rng.normal_(buf)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            rng.normal_(buf)
        g.replay(); s.synchronize(); first = buf.clone()
        g.replay(); s.synchronize()
    assert not torch.equal(first, buf)
    c = rnd((3, 64, 128), 41, dev, torch.bfloat16)
    assert torch.equal(ops.slice_cast(c, 16, 10), c[:, :16, :10].float())
    f = rnd((5, 37, 24), 42, dev)
    assert torch.equal(ops.cast(f, torch.bfloat16), f.to(torch.bfloat16))
    assert torch.equal(ops.cast(f.to(torch.bfloat16), torch.float32), f.to(torch.bfloat16).float())
