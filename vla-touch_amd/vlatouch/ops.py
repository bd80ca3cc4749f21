"""Thin tensor-level wrappers over the primitive C-ABI entries (vt_gemm, vt_attention, vt_groupnorm,
vt_rownorm).  Used by the parity tests and by host code that composes primitives directly."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _es(t: torch.Tensor) -> int:
    return t.element_size()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = L.ACT_NONE,
         colscale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, splitk: int = 1,
         headnorm=None, cmap=None, wp: Optional[torch.Tensor] = None, sk_ws: Optional[torch.Tensor] = None,
         sk_cnt: Optional[torch.Tensor] = None, xn=None, rs=None) -> torch.Tensor:
    """out[M,N] = residual + colscale * act(a[M,K] @ w[N,K]^T + bias).  splitk>1 returns the fp32 slabs [splitk,M,N].
    headnorm = (w0[64], c0_end, w1[64] | None, c1_end, eps, mode): fused per-head RMSNorm (large bf16 GEMMs only).
    cmap = (mode, T) with out= a [H, T, 2, 64, 64] tile stream (T = ceil(M/64)): the cached-condition K (mode 1) / Vt (mode 2) layout.
    RMSNorm hand-off between two Linears on the weights-in-registers tile (csrc/vt_gemm.h): xn = (xn_out [M, N] 16-bit, gain [N] fp32, part [M, 2N/128, 2] fp32:
    (sum of squares, sum) per 64 columns — with mode = NORM_RMS_VAR as a 4th entry: (second moment about the 64 columns' own mean, sum)) on the residual Linear (fp32 out), rs = (part, eps[, mode = NORM_RMS_MEANSQ | NORM_RMS_VAR]) on the Linear that reads
    xn_out as `a`."""
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    out_dtype = out_dtype or (torch.float32 if splitk > 1 else a.dtype)
    if out is None:
        out = torch.empty((splitk, M, N) if splitk > 1 else (M, N), dtype=out_dtype, device=a.device)
    p = L.GemmParams()
    p.A, p.W, p.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldw, p.ldc = a.stride(0), w.stride(0), N
    p.bias = 0 if bias is None else bias.data_ptr()
    p.colscale = 0 if colscale is None else colscale.data_ptr()
    if residual is not None:
        assert residual.dtype == out_dtype
        p.residual, p.ldr = residual.data_ptr(), residual.stride(0)
    p.act, p.groups, p.splitk = act, 1, splitk
    p.c_slab = M * N
    p.a_dtype, p.w_dtype, p.c_dtype = L.dt_code(a.dtype), L.dt_code(w.dtype), L.dt_code(out_dtype)
    if headnorm is not None:
        w0, c0, w1, c1, eps, mode = headnorm
        p.hn_w0, p.hn_c0_end = w0.data_ptr(), c0
        p.hn_w1, p.hn_c1_end = (0 if w1 is None else w1.data_ptr()), c1
        p.hn_eps, p.hn_mode = eps, mode
    if cmap is not None:
        p.cmap, p.cmap_T = cmap
    if wp is not None:                       # fragment-packed copy of w (pack_w32): lets the dispatcher pick the weights-in-registers tile
        assert wp.numel() == w.numel() and wp.dtype == w.dtype
        p.Wp = wp.data_ptr()
    if sk_ws is not None and sk_cnt is not None:      # split-K scratch of the small-M tile: fp32 slab area + zeroed int32 ticket counters
        assert sk_cnt.dtype == torch.int32
        p.sk_ws, p.sk_ws_bytes, p.sk_cnt, p.sk_cnt_n = sk_ws.data_ptr(), sk_ws.numel() * sk_ws.element_size(), sk_cnt.data_ptr(), sk_cnt.numel()
    if xn is not None:
        xo, gain, part = xn[:3]
        p.rs_mode = xn[3] if len(xn) > 3 else L.NORM_RMS_MEANSQ      # the variance form hands over CENTRED second moments per 64 columns (csrc/vt_gemm.h)
        assert xo.shape == (M, N) and xo.dtype == a.dtype and gain.dtype == torch.float32 and part.dtype == torch.float32 and part.shape == (M, 2 * N // 128, 2) and part.is_contiguous()
        p.xn_out, p.xn_ld, p.xn_gain, p.xn_part = xo.data_ptr(), xo.stride(0), gain.data_ptr(), part.data_ptr()
    if rs is not None:
        part, eps = rs[0], rs[1]
        assert part.dtype == torch.float32 and part.shape[0] == M and part.dim() == 3 and part.shape[2] == 2 and part.is_contiguous()
        p.rs_part, p.rs_n, p.rs_inv_k, p.rs_eps = part.data_ptr(), part.shape[1], 1.0 / K, eps
        p.rs_mode = rs[2] if len(rs) > 2 else L.NORM_RMS_MEANSQ
    L.check(L.lib().vt_gemm(C.byref(p), L.stream_ptr(a.device)), "vt_gemm")
    return out


class DeviceRng:
    """Philox key + counter in device memory (vt_randn): N(0,1) draws without torch kernels, graph-replay safe."""

    def __init__(self, seed: int, device="cuda"):
        self.device = L.require_gpu(device)
        self.state = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=self.device)

    def normal_(self, out: torch.Tensor, round_bf16: bool = False) -> torch.Tensor:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.device.type == "cuda"
        L.check(L.lib().vt_randn(L.ptr(out), out.numel(), L.ptr(self.state), int(round_bf16), L.stream_ptr(out.device)), "vt_randn")
        return out


def slice_cast(x: torch.Tensor, T: int, D: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B, Tin, Din] (bf16 / fp32, contiguous) -> fp32 [B, T, D] = x[:, :T, :D].float() in one kernel."""
    assert x.dim() == 3 and x.is_contiguous()
    B, Tin, Din = x.shape
    if out is None:
        out = torch.empty(B, T, D, dtype=torch.float32, device=x.device)
    L.check(L.lib().vt_slice_cast(L.ptr(x), L.dt_code(x.dtype), L.ptr(out), B, Tin, Din, T, D, L.stream_ptr(x.device)), "vt_slice_cast")
    return out


def cast(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x.to(dtype) through vt_cast (one library kernel, no torch kernel): x contiguous, any shape."""
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    cols = x.shape[-1]
    rows = x.numel() // cols
    L.check(L.lib().vt_cast(L.ptr(x), L.dt_code(x.dtype), cols, L.ptr(out), L.dt_code(dtype), cols, rows, cols, L.stream_ptr(x.device)), "vt_cast")
    return out


def pack_w32(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] 16-bit (N % 32 == 0, K % 16 == 0) -> its copy in MFMA fragment order (vt_pack_w32)."""
    assert w.dim() == 2 and w.stride(1) == 1 and w.element_size() == 2
    out = torch.empty(w.numel(), dtype=w.dtype, device=w.device)
    L.check(L.lib().vt_pack_w32(L.ptr(w), w.stride(0), L.ptr(out), w.shape[0], w.shape[1], L.stream_ptr(w.device)), "vt_pack_w32")
    return out


def conv1d_cl(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, taps: int, cin: int, tout: int,
              stride: int = 1, off0: int = 0, tstep: int = 1, out_dtype: Optional[torch.dtype] = None, splitk: int = 1) -> torch.Tensor:
    """Channel-last implicit-GEMM conv: x [B, Tin, cin] , w_packed [Cout, taps*cin] -> [B, tout, Cout]."""
    B, tin, c = x.shape
    assert c == cin and x.is_contiguous()
    N = w_packed.shape[0]
    out_dtype = out_dtype or (torch.float32 if splitk > 1 else x.dtype)
    out = torch.empty((splitk, B, tout, N) if splitk > 1 else (B, tout, N), dtype=out_dtype, device=x.device)
    p = L.GemmParams()
    p.A, p.W, p.C = x.data_ptr(), w_packed.data_ptr(), out.data_ptr()
    p.M, p.N, p.K = B * tout, N, taps * cin
    p.lda, p.ldw, p.ldc = cin, taps * cin, N
    p.taps, p.cin, p.tout, p.tin, p.stride, p.off0, p.tstep = taps, cin, tout, tin, stride, off0, tstep
    p.bias = 0 if bias is None else bias.data_ptr()
    p.groups, p.splitk = 1, splitk
    p.c_slab = B * tout * N
    p.a_dtype, p.w_dtype, p.c_dtype = L.dt_code(x.dtype), L.dt_code(w_packed.dtype), L.dt_code(out_dtype)
    L.check(L.lib().vt_gemm(C.byref(p), L.stream_ptr(x.device)), "vt_gemm(conv)")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, scale: Optional[float] = None,
              kmask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [B,Nq,H,hd], k/v [B,Nk,H,hd], hd 64, 96 or (16-bit, unmasked) 80 (any strides with unit inner stride) -> [B,Nq,H*hd].  kmask [B,Nk] bool."""
    B, Nq, H, hd = q.shape
    Nk = k.shape[1]
    assert hd in (64, 80, 96) and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    o = torch.empty(B, Nq, H * hd, dtype=q.dtype, device=q.device)
    p = L.AttnParams()
    p.Q, p.K, p.V, p.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    p.q_bs, p.q_rs, p.q_hs = q.stride(0), q.stride(1), q.stride(2)
    p.k_bs, p.k_rs, p.k_hs = k.stride(0), k.stride(1), k.stride(2)
    p.v_bs, p.v_rs, p.v_hs = v.stride(0), v.stride(1), v.stride(2)
    p.o_bs, p.o_rs = Nq * H * hd, H * hd
    p.hd = hd
    km = None
    if kmask is not None:
        km = kmask.to(torch.uint8).contiguous()
        p.kmask, p.km_bs = km.data_ptr(), Nk
    p.B, p.H, p.Nq, p.Nk = B, H, Nq, Nk
    p.scale = scale if scale is not None else hd ** -0.5
    p.dtype = L.dt_code(q.dtype)
    L.check(L.lib().vt_attention(C.byref(p), L.stream_ptr(q.device)), "vt_attention")
    return o


def rownorm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], eps: float, mode: int = L.NORM_LAYER,
            out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    assert x.dim() == 2 and x.stride(1) == 1
    out_dtype = out_dtype or x.dtype
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    L.check(L.lib().vt_rownorm(L.ptr(x), L.dt_code(x.dtype), x.stride(0), L.ptr(y), L.dt_code(out_dtype), y.stride(0), L.ptr(w), L.ptr(b),
                               x.shape[0], x.shape[1], eps, mode, L.stream_ptr(x.device)), "vt_rownorm")
    return y


def headnorm_(x: torch.Tensor, heads: int, w: torch.Tensor, eps: float = 1e-6, mode: int = L.NORM_RMS_MEANSQ, tok_stride: Optional[int] = None,
              tokens: Optional[int] = None) -> None:
    """In-place per-head RMSNorm over 64-wide head slices of a [tokens, >=heads*64] buffer (q_norm / k_norm)."""
    tok_stride = tok_stride if tok_stride is not None else x.stride(-2)
    tokens = tokens if tokens is not None else x.numel() // x.shape[-1]
    L.check(L.lib().vt_headnorm(L.ptr(x), L.dt_code(x.dtype), tok_stride, heads, tokens, L.ptr(w), eps, mode, L.stream_ptr(x.device)), "vt_headnorm")


def groupnorm_cl(slabs: torch.Tensor, bias, gamma, beta, *, B: int, T: int, ngroups: int = 8, film: Optional[torch.Tensor] = None,
                 residual: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float32, eps: float = 1e-5) -> torch.Tensor:
    """slabs [S, B*T, C] fp32 partial sums -> GroupNorm -> Mish -> FiLM(film [B, 2C]) -> + residual [B*T, C]."""
    S, M, Cc = slabs.shape
    assert M == B * T
    out = torch.empty(M, Cc, dtype=out_dtype, device=slabs.device)
    p = L.GnParams()
    p.P, p.nslabs, p.slab_stride, p.p_gs, p.ldp = slabs.data_ptr(), S, M * Cc, 0, Cc
    p.bias = 0 if bias is None else bias.data_ptr()
    p.gamma, p.beta, p.vec_gs = gamma.data_ptr(), beta.data_ptr(), 0
    if film is not None:
        p.film, p.film_ld, p.film_off, p.film_gs = film.data_ptr(), film.stride(0), 0, 0
    if residual is not None:
        assert residual.dtype == out_dtype
        p.residual, p.ldr, p.r_gs = residual.data_ptr(), residual.stride(0), 0
    p.out, p.ldo, p.o_gs, p.out_dtype = out.data_ptr(), Cc, 0, L.dt_code(out_dtype)
    p.B, p.T, p.C, p.ngroups, p.nets, p.eps = B, T, Cc, ngroups, 1, eps
    L.check(L.lib().vt_groupnorm(C.byref(p), L.stream_ptr(slabs.device)), "vt_groupnorm")
    return out


def mfma_selftest(device="cuda"):
    err = torch.full((2,), -1.0, dtype=torch.float32, device=device)
    L.check(L.lib().vt_selftest_mfma(L.ptr(err), L.stream_ptr(err.device)), "vt_selftest_mfma")
    return err.cpu().tolist()
