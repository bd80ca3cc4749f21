"""The persistent 256-square GEMM tile with the in-loop, in-register epilogue (csrc/vt_gemm_pt.hip) against (a) a torch fp32 reference and
(b) gemm_pp256d_kernel on the same launch (`vt_tune(8, 0)`): row-major 16-bit outputs must be BIT-identical (same accumulation order, same
epilogue arithmetic), the fused K|V projection's Vt half too; its K half sums the 64 squares of a head row in another order (16 per lane, then
across 4 lanes), so it may differ from the old kernel by one 16-bit rounding step.  Shapes cover: one tile per block (flat epilogue only), many
tiles per block (slots inside the next tile's first k-tile), odd k-tile counts (buffer parity flips per tile), ragged M (row range check, V-half
zero fill), N that is not a multiple of the tile (whole waves out of range)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from vlatouch import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rnd(shape, seed, dev, dtype=torch.float32, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def both(fn):
    """fn() with the persistent tile on and off."""
    from vlatouch import _lib as L
    lib = L.lib()
    try:
        L.check(lib.vt_tune(8, 1), "tune")
        new = fn()
        torch.cuda.synchronize()
        L.check(lib.vt_tune(8, 0), "tune")
        old = fn()
        torch.cuda.synchronize()
    finally:
        lib.vt_tune(8, 1)
    return new, old


def gelu_ref(y, act):
    from vlatouch import _lib as L
    if act == L.ACT_GELU_ERF:
        return torch.nn.functional.gelu(y)
    if act == L.ACT_GELU_TANH:
        return torch.nn.functional.gelu(y, approximate="tanh")
    return y


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,act", [
    (16384, 2304, 768, "none"),       # DINOv2-B qkv: 576 tiles, 2.25 per block, N = 9 tiles
    (16500, 3072, 768, "erf"),        # fc1 + GELU, ragged M (116 rows in the last m-tile)
    (36000, 1152, 1152, "none"),      # SigLIP qkv-like: N = 4.5 tiles (two waves of the last n-tile out of range), nk = 18
    (35000, 2048, 576, "tanh"),       # nk = 9: the LDS buffer parity flips from tile to tile
    (2144, 6144, 2048, "none"),       # 216 tiles on 216 blocks: one tile per block, epilogue entirely in the flat tail
    (70001, 1024, 512, "erf"),        # M % 4 == 1, 4 n-tiles, 4.3 tiles per block
    (16448, 768, 1024, "tanh"),       # 195 tiles on 195 blocks (not a multiple of 8: plain tile order)
    (65537, 2112, 512, "erf"),        # the smallest K the tile takes (8 k-tiles), M % 256 == 1 (one live row in the last m-tile), N % 256 == 64 (one wave column live)
    (33023, 1088, 512, "tanh"),       # M % 256 == 255 (one dead row), N % 256 == 64, K = 512
    (33023, 2112, 576, "none"),       # the same edges at an odd k-tile count
])
def test_persistent_tile_row_major_16bit(dev, dtype, M, N, K, act):
    from vlatouch import ops, _lib as L
    code = {"none": L.ACT_NONE, "erf": L.ACT_GELU_ERF, "tanh": L.ACT_GELU_TANH}[act]
    a = rnd((M, K), 1, dev, dtype)
    w = rnd((N, K), 2, dev, dtype, K ** -0.5)
    bias = rnd((N,), 3, dev)
    out = torch.full((M + 3, N), 7.0, dtype=dtype, device=dev)          # 3 guard rows: nothing may be written past row M

    def run():
        out.fill_(7.0)
        ops.gemm(a, w, bias, act=code, out=out[:M], out_dtype=dtype)
        return out.clone()
    new, old = both(run)
    assert torch.equal(new[M:], torch.full((3, N), 7.0, dtype=dtype, device=dev))
    assert torch.equal(new, old), float((new.float() - old.float()).abs().max())
    ref = gelu_ref(a.float() @ w.float().t() + bias, code)
    err = float((new[:M].float() - ref).abs().max() / ref.abs().max())
    assert err < (1e-2 if dtype == torch.bfloat16 else 2e-3), err
    new2, _ = both(run)
    assert torch.equal(new, new2)                                        # deterministic


def _kv_reference(a, w, bias, gain, mode, M, T):
    """fp32 K|V projection -> ([H, T*64, 64] K after the per-head RMSNorm, [H, T*64, 64] V), rows >= M zero."""
    from vlatouch import _lib as L
    N = w.shape[0]
    D, H = N // 2, N // 128
    y = a.float() @ w.float().t() + bias
    k = y[:, :D].reshape(M, H, 64)
    var = k.pow(2).mean(-1, keepdim=True) if mode == L.NORM_RMS_MEANSQ else k.var(-1, keepdim=True)
    k = k * torch.rsqrt(var + 1e-6) * gain
    v = y[:, D:].reshape(M, H, 64)
    pad = lambda t: torch.cat([t, torch.zeros(T * 64 - M, H, 64, device=t.device)]).permute(1, 0, 2)
    return pad(k), pad(v)


def _untile(kv, T):
    """tile stream [H, T, 2, 64, 64] -> K [H, T*64, 64], V [H, T*64, 64] (undoing the Vt tiles' MFMA key order)."""
    H = kv.shape[0]
    kk = torch.arange(64, device=kv.device)
    pos = (kk & 32) | (((kk >> 2) & 3) << 3) | (((kk >> 4) & 1) << 2) | (kk & 3)      # vt_kpos
    k = kv[:, :, 0].reshape(H, T * 64, 64)
    v = kv[:, :, 1][:, :, :, pos].permute(0, 1, 3, 2).reshape(H, T * 64, 64)            # [h, t, d, kk] -> [h, t, kk, d]
    return k, v


@pytest.mark.parametrize("M,K,mode", [(70001, 512, "meansq"), (13122, 2048, "var"), (139968 // 4, 2048, "meansq"), (4374, 2048, "meansq"),
                                      (65537, 512, "var"), (33023, 512, "meansq")])      # M % 256 in {1, 255} at the smallest K
@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16])      # fp16: the activation type of the default RDT mode (round 5)
def test_persistent_tile_fused_kv_projection(dev, M, K, mode, dt16):
    from vlatouch import ops, _lib as L
    N, H = 4096, 32
    T = (M + 63) // 64
    m = L.NORM_RMS_MEANSQ if mode == "meansq" else L.NORM_RMS_VAR
    a = rnd((M, K), 1, dev, dt16)
    w = rnd((N, K), 2, dev, dt16, K ** -0.5)
    bias = rnd((N,), 3, dev)
    gain = rnd((64,), 4, dev) * 0.2 + 1.0
    # the stream the kernel addresses is [H][T] tile pairs back to back: use an exactly sized buffer and a guard behind it
    flat = torch.empty(H * T * 2 * 4096 + 8192, dtype=dt16, device=dev)

    def kv_view():
        return flat[: H * T * 2 * 4096].view(H, T, 2, 64, 64)

    def run2():
        flat.fill_(7.0)
        ops.gemm(a, w, bias, out=kv_view(), out_dtype=dt16, headnorm=(gain, N // 2, None, N // 2, 1e-6, m), cmap=(3, T))
        return flat.clone()
    new, old = both(run2)
    assert torch.equal(new[H * T * 2 * 4096:], torch.full((8192,), 7.0, dtype=dt16, device=dev))      # nothing past the stream
    kn, vn = _untile(new[: H * T * 2 * 4096].view(H, T, 2, 64, 64), T)
    ko, vo = _untile(old[: H * T * 2 * 4096].view(H, T, 2, 64, 64), T)
    assert torch.equal(vn[:, :M], vo[:, :M])                                             # V half: bit-identical to the old kernel
    tail = vn[:, M:]
    assert bool(((tail == 0) | (tail == 7.0)).all())                                     # rows >= M: zero-filled or untouched, never garbage
    kd = (kn[:, :M].float() - ko[:, :M].float()).abs()
    assert float(kd.max()) <= (2.0 ** -6 if dt16 == torch.bfloat16 else 2.0 ** -9) * float(ko[:, :M].float().abs().max())          # K half: at most a rounding step apart
    assert float((kd > 0).float().mean()) < 0.02
    assert bool((kn[:, M:] == 7.0).all())                                                # K rows >= M are not written
    kr, vr = _kv_reference(a, w, bias, gain, m, M, T)
    tol = 1e-2 if dt16 is torch.bfloat16 else 2e-3
    assert float((kn[:, :M].float() - kr[:, :M]).abs().max() / kr.abs().max()) < tol
    assert float((vn[:, :M].float() - vr[:, :M]).abs().max() / vr.abs().max()) < tol
    again, _ = both(run2)
    assert torch.equal(new, again)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,inplace", [
    (139968 // 2, 1152, 1152, True),      # SigLIP out-projection shape (half the batch): 4.5 n-tiles, 5 tiles per block, residual stream updated in place
    (16384, 768, 3072, False),            # DINOv2-B fc2: 192 tiles, one per block (flat epilogue only), separate output
    (40000, 768, 768, True),              # ragged M, 3 n-tiles, nk = 12
    (33000, 2048, 576, False),            # nk = 9 (buffer parity flips per tile)
    (16448, 768, 3072, True),             # 65 x 3 = 195 tiles: one block per tile, a grid that is not a multiple of 8 (plain tile order), ragged last m-tile
    (16448, 768, 768, True),              # DINOv2-B out-projection: one round at K = 768, a shape only the persistent kernel takes (vt_gemm_pt_extra_shape)
    (65537, 1088, 512, True),             # smallest K, M % 256 == 1, N % 256 == 64
    (33023, 2112, 512, False),            # M % 256 == 255, N % 256 == 64
])
def test_persistent_tile_fp32_residual_kind(dev, dtype, M, N, K, inplace):
    """R32 kind: C (fp32) = residual + colscale * (A W^T + bias) — the LayerScale + residual epilogue of the ViT out-projection / fc2 — against
    gemm_pp256d_kernel on the same launch and a torch fp32 reference."""
    from vlatouch import ops
    a = rnd((M, K), 1, dev, dtype)
    w = rnd((N, K), 2, dev, dtype, K ** -0.5)
    bias = rnd((N,), 3, dev)
    cs = rnd((N,), 4, dev) * 0.3 + 1.0
    res0 = rnd((M + 2, N), 5, dev)                                        # 2 guard rows
    buf = torch.empty_like(res0)

    def run():
        buf.copy_(res0)
        if inplace:
            ops.gemm(a, w, bias, colscale=cs, residual=buf[:M], out=buf[:M], out_dtype=torch.float32)
            return buf.clone()
        out = torch.full((M + 2, N), 7.0, device=dev)
        ops.gemm(a, w, bias, colscale=cs, residual=buf[:M], out=out[:M], out_dtype=torch.float32)
        return torch.cat([out, buf])
    new, old = both(run)
    guard = res0[M:] if inplace else torch.full((2, N), 7.0, device=dev)
    assert torch.equal(new[M:M + 2], guard)                               # nothing written past row M
    scale = float(old[:M].abs().max())
    assert float((new - old).abs().max()) <= 2e-6 * scale                 # same arithmetic (the compiler may contract mul + add differently)
    ref = res0[:M] + cs * (a.float() @ w.float().t() + bias)
    err = float((new[:M] - ref).abs().max() / ref.abs().max())
    assert err < (1e-2 if dtype == torch.bfloat16 else 2e-3), err
    again, _ = both(run)
    assert torch.equal(new, again)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode,shift", [("meansq", 2.0), ("var", 2.0), ("var", 400.0)])
@pytest.mark.parametrize("M,N2,hn", [(2144, 6144, True), (2144, 2048, False), (2100, 2048, True)])
def test_rmsnorm_handoff_between_two_linears(dev, dtype, M, N2, hn, mode, shift):
    """The fused RMSNorm hand-off of the per-denoise-step Linears (vt_gemm.h xn_out / rs_part): residual Linear -> [norm] -> Linear with the norm launch
    replaced by (x * gain, (sum of squares, sum) per 64 columns) out of the first epilogue and a row scale in the second, against the three-launch form and
    torch fp32 — in both RmsNorm forms: mean-square (timm >= 1.0.9) and unbiased variance of the un-centred row (timm <= 1.0.8 incl. the timm==1.0.3
    upstream RDT-1B pins, models/rdt/blocks.py:22,150-156).  The stream carries a row offset (mean ~ 0.6 of the rms) so that the two forms differ; shift = 400
    (round 6): |row mean| = 130 x the standard deviation, where sum x^2 - (sum x)^2 / K has lost 4 of fp32's 7 digits — the variance form hands over
    centred second moments per 64 columns and merges them pairwise, as accurate as the two-pass row-norm kernel of the three-launch form."""
    from vlatouch import ops, _lib as L
    D = 2048
    m = L.NORM_RMS_MEANSQ if mode == "meansq" else L.NORM_RMS_VAR
    a = rnd((M, D), 1, dev, dtype)
    w1 = rnd((D, D), 2, dev, dtype, D ** -0.5)
    b1 = rnd((D,), 3, dev)
    x0 = rnd((M, D), 4, dev) * 3.0 + shift
    gain = rnd((D,), 5, dev) * 0.2 + 1.0
    w2 = rnd((N2, D), 6, dev, dtype, D ** -0.5)
    b2 = rnd((N2,), 7, dev)
    hw = (rnd((64,), 8, dev) * 0.1 + 1.0) if hn else None
    wp1, wp2 = ops.pack_w32(w1), ops.pack_w32(w2)
    head = (hw, N2, None, N2, 1e-6, m) if hn else None
    act = L.ACT_NONE if hn else L.ACT_GELU_TANH
    # three launches
    xa = x0.clone()
    ops.gemm(a, w1, b1, residual=xa, out=xa, out_dtype=torch.float32, wp=wp1)
    xn_ref = ops.rownorm(xa, gain, None, eps=1e-6, mode=m, out_dtype=dtype)
    y3 = ops.gemm(xn_ref, w2, b2, act=act, headnorm=head, wp=wp2)
    # hand-off
    xb = torch.cat([x0, torch.full((2, D), 7.0, device=dev)])                  # guard rows
    xo = torch.full((M + 2, D), 7.0, dtype=dtype, device=dev)
    part = torch.full((M + 2, 2 * D // 128, 2), 7.0, device=dev)
    ops.gemm(a, w1, b1, residual=xb[:M], out=xb[:M], out_dtype=torch.float32, wp=wp1, xn=(xo[:M], gain, part[:M], m))
    assert torch.equal(xb[:M], xa)                                             # the fp32 stream itself: the same arithmetic
    assert bool((xb[M:] == 7.0).all()) and bool((xo[M:] == 7.0).all()) and bool((part[M:] == 7.0).all())
    grp = xa.double().view(M, -1, 64)
    sm = grp.sum(-1)
    sq = grp.pow(2).sum(-1) if mode == "meansq" else (grp - grp.mean(-1, keepdim=True)).pow(2).sum(-1)      # variance form: about the 64 columns' own mean
    assert float((part[:M, :, 0].double() - sq).abs().max() / sq.max()) < 1e-5
    assert float((part[:M, :, 1].double() - sm).abs().max() / sm.abs().max()) < 1e-5
    y2 = ops.gemm(xo[:M], w2, b2, act=act, headnorm=head, wp=wp2, rs=(part[:M].contiguous(), 1e-6, m))
    # torch fp32 reference from the same fp32 stream
    stat = xa.pow(2).mean(-1, keepdim=True) if mode == "meansq" else xa.var(-1, keepdim=True)
    xr = xa * torch.rsqrt(stat + 1e-6) * gain
    ref = xr @ w2.float().t() + b2
    if hn:
        r = ref.view(M, -1, 64)
        hstat = r.pow(2).mean(-1, keepdim=True) if mode == "meansq" else r.var(-1, keepdim=True)
        ref = (r * torch.rsqrt(hstat + 1e-6) * hw).reshape(M, N2)
    else:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    scale = float(ref.abs().max())
    e2, e3 = float((y2.float() - ref).abs().max()) / scale, float((y3.float() - ref).abs().max()) / scale
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert e2 < tol and e2 < 1.5 * e3 + 1e-4, (e2, e3)                         # as close to fp32 as the three-launch form
    y2b = ops.gemm(xo[:M], w2, b2, act=act, headnorm=head, wp=wp2, rs=(part[:M].contiguous(), 1e-6, m))
    assert torch.equal(y2, y2b)
    if mode == "var":      # and the two forms really differ on this stream (a consumer that ignored rs_mode would pass the mean-square reference)
        xc, part_ms = x0.clone(), torch.empty_like(part[:M])
        ops.gemm(a, w1, b1, residual=xc, out=xc, out_dtype=torch.float32, wp=wp1, xn=(xo[:M], gain, part_ms, L.NORM_RMS_MEANSQ))
        y_ms = ops.gemm(xo[:M], w2, b2, act=act, headnorm=head, wp=wp2, rs=(part_ms, 1e-6, L.NORM_RMS_MEANSQ))
        if not hn:
            assert float((y_ms.float() - y2.float()).abs().max()) / scale > 2e-2
