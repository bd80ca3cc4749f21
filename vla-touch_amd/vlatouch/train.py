"""Training step of the interpolant controller on the MI355X.

Replaces, for one optimisation step, what the reference does with torch autograd:
  * `StochasticInterpolants.get_loss` = velocity_loss + score_loss + b_loss over the three conditional 1-D U-Nets
    (VLA/residual_controller/bridge/bridge_model.py:183-258) — q_sample, the three forwards, the three losses;
  * `total_loss.backward()` through the U-Nets (bridge/networks/conditional_unet_1D.py:58-247) and, through `obs_cond`, the
    state/force observation MLP (bridge_controller.py:42-48; the DINOv2 features are inputs, frozen);
  * `optim.AdamW(net.parameters() + state_encoder.parameters()).step()` and `ema.update()` (bridge_train.py:49-58, 312-334).

Design: fp32 master weights live on the device in the PACKED layouts the kernels consume (conv weights tap-major [Cout][k*Cin],
ConvTranspose1d as its equivalent stride-1 convolution over a zero-stuffed input, the 12 FiLM Linears concatenated), so gradients and
optimizer state are in the same layouts and nothing is re-laid-out per step except the transposed / flipped copies a data gradient
needs.  Every matrix product — forward convolutions, weight gradients (dY^T x im2col(X)^T), data gradients (a convolution of dY with
the flipped weights) — is a vt_gemm call in exact-fp32 MFMA mode; GroupNorm+Mish+FiLM forward is the fused inference kernel, its
backward / the losses / AdamW / EMA are csrc/vt_train.hip.  The orchestration below is host Python like the reference's training
loop: a training step is a few thousand small launches (this is the f-4 row "built and parity-checked", not a tuned trainer).
`state_dict()` / `load_state_dict()` speak the reference's key names and tensor layouts (checkpoint compatible).
"""
from __future__ import annotations

import os

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops

F32 = torch.float32


def _sp(dev):
    return L.stream_ptr(dev)


def _empty(shape, dev):
    return torch.empty(shape, dtype=F32, device=dev)


# ---------------------------------------------------------------------------------------------- primitive wrappers
def transpose(x2d: torch.Tensor) -> torch.Tensor:
    M, N = x2d.shape
    out = _empty((N, M), x2d.device)
    L.check(L.lib().vt_transpose(L.ptr(x2d), L.ptr(out), M, N, _sp(x2d.device)), "vt_transpose")
    return out


def im2col_t(x: torch.Tensor, tout: int, taps: int, stride: int, off0: int) -> torch.Tensor:
    B, tin, cin = x.shape
    out = _empty((taps * cin, B * tout), x.device)
    L.check(L.lib().vt_im2col_t(L.ptr(x), L.ptr(out), B, tin, tout, cin, taps, stride, off0, _sp(x.device)), "vt_im2col_t")
    return out


def zero_stuff(x: torch.Tensor) -> torch.Tensor:
    B, T, Cc = x.shape
    out = _empty((B, 2 * T, Cc), x.device)
    L.check(L.lib().vt_zero_stuff(L.ptr(x), L.ptr(out), B, T, Cc, _sp(x.device)), "vt_zero_stuff")
    return out


def wflip(wp: torch.Tensor, cout: int, taps: int, cin: int) -> torch.Tensor:
    out = _empty((cin, taps * cout), wp.device)
    L.check(L.lib().vt_wflip(L.ptr(wp), L.ptr(out), cout, taps, cin, _sp(wp.device)), "vt_wflip")
    return out


def colsum(x2d: torch.Tensor) -> torch.Tensor:
    M, N = x2d.shape
    out = _empty((N,), x2d.device)
    L.check(L.lib().vt_colsum(L.ptr(x2d), x2d.stride(0), L.ptr(out), M, N, 0, _sp(x2d.device)), "vt_colsum")
    return out


def add_(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    L.check(L.lib().vt_add_(L.ptr(a), L.ptr(b), a.numel(), _sp(a.device)), "vt_add_")
    return a


def copy_cols(src: torch.Tensor, off: int, dst: torch.Tensor, doff: int, cols: int, accumulate: bool = False) -> None:
    """dst[:, doff:doff+cols] (+)= src[:, off:off+cols] on 2-D views of row-contiguous buffers."""
    rows = src.shape[0]
    L.check(L.lib().vt_copy_cols(L.ptr(src), src.stride(0), off, L.ptr(dst), dst.stride(0), doff, rows, cols, int(accumulate), _sp(src.device)), "vt_copy_cols")


def mish(x: torch.Tensor, dy: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(x)
    L.check(L.lib().vt_mish(L.ptr(x), L.ptr(dy), L.ptr(out), x.numel(), _sp(x.device)), "vt_mish")
    return out


def gelu(x: torch.Tensor, dy: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(x)
    L.check(L.lib().vt_gelu(L.ptr(x), L.ptr(dy), L.ptr(out), x.numel(), _sp(x.device)), "vt_gelu")
    return out


def _splits(M: int, N: int, K: int) -> int:
    """Split-K factor for the fp32 64x64-tile GEMM: the backward pass is full of products with few output tiles and a long reduction
    (weight gradients: [Cout, k*Cin] outputs reduced over B*T rows; the T=4 levels: 512-row activations against k*Cin = 2560..5120) —
    one 4-wave block per tile leaves most of the 256 CUs idle.  Aim at >= 512 blocks of 64x64 (two per CU, what the LDS ring of
    csrc/vt_gemm_f32r.hip is sized for) while a slice keeps >= 256 of the reduction (tools/gemm_bench_f32.py: within 5 % of the best
    factor on every shape of the step; more splits only add slab traffic)."""
    if N % 4 or os.environ.get("VLATOUCH_TRAIN_SPLITK", "1") == "0":
        return 1
    blocks = ((M + 63) // 64) * ((N + 63) // 64)
    s = 1
    while s < 16 and blocks * s < 512 and K // (2 * s) >= 256:
        s *= 2
    return s


def _slab_sum(slabs: torch.Tensor, bias, shape) -> torch.Tensor:
    S = slabs.shape[0]
    out = _empty(shape, slabs.device)
    L.check(L.lib().vt_slab_sum(L.ptr(slabs), S, out.numel(), L.ptr(bias), shape[-1], L.ptr(out), _sp(slabs.device)), "vt_slab_sum")
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    """a [M,K] x w [N,K]^T (+ bias) in fp32, split along K when the output has too few tiles to fill the chip (deterministic reduce)."""
    M, K = a.shape
    N = w.shape[0]
    s = _splits(M, N, K)
    if s == 1:
        return ops.gemm(a, w, bias)
    return _slab_sum(ops.gemm(a, w, None, splitk=s), bias, (M, N))


def conv(x, wp, b, *, taps, cin, tout, stride, off0):
    B = x.shape[0]
    N = wp.shape[0]
    s = _splits(B * tout, N, taps * cin)
    if s == 1:
        return ops.conv1d_cl(x, wp, b, taps=taps, cin=cin, tout=tout, stride=stride, off0=off0)
    return _slab_sum(ops.conv1d_cl(x, wp, None, taps=taps, cin=cin, tout=tout, stride=stride, off0=off0, splitk=s), b, (B, tout, N))


def conv_fwd(x, wp, b, k, stride, pad, tout):
    return conv(x, wp, b, taps=k, cin=x.shape[2], tout=tout, stride=stride, off0=-pad)


def conv_bwd(x, wp, dy, k, stride, pad):
    """x [B,Tin,Cin], wp [Cout, k*Cin], dy [B,Tout,Cout] -> dx [B,Tin,Cin], dwp [Cout,k*Cin], db [Cout]."""
    B, tin, cin = x.shape
    tout, cout = dy.shape[1], dy.shape[2]
    dy2 = dy.reshape(B * tout, cout)
    dwp = gemm(transpose(dy2), im2col_t(x, tout, k, stride, -pad))            # [Cout, M] x [k*Cin, M]^T
    db = colsum(dy2)
    wt = wflip(wp, cout, k, cin)                                                  # [Cin, k*Cout], taps reversed
    src = dy if stride == 1 else zero_stuff(dy)                                   # strided conv: its data gradient is a transposed conv
    dx = conv(src, wt, None, taps=k, cin=cout, tout=tin, stride=1, off0=pad - (k - 1))
    return dx, dwp, db


def convT_fwd(x, wc, b):
    """ConvTranspose1d(k=4, s=2, p=1) as the stride-1 conv of the zero-stuffed input with wc[co][tp][ci] = W[ci][co][3 - tp]."""
    xz = zero_stuff(x)
    return conv(xz, wc, b, taps=4, cin=x.shape[2], tout=2 * x.shape[1], stride=1, off0=-2), xz


def convT_bwd(xz, wc, dy, cin):
    B, t2, cout = dy.shape
    dy2 = dy.reshape(B * t2, cout)
    dwc = gemm(transpose(dy2), im2col_t(xz, t2, 4, 1, -2))
    db = colsum(dy2)
    wt = wflip(wc, cout, 4, cin)
    dx = conv(dy, wt, None, taps=4, cin=cout, tout=t2 // 2, stride=2, off0=-1)    # d xz at the even (non-stuffed) positions
    return dx, dwc, db


def gn_fwd(c, gamma, beta, film=None, residual=None):
    B, T, Cc = c.shape
    out = ops.groupnorm_cl(c.reshape(1, B * T, Cc), None, gamma, beta, B=B, T=T, ngroups=8, film=film,
                           residual=None if residual is None else residual.reshape(B * T, Cc))
    return out.reshape(B, T, Cc)


def gn_bwd(c, gamma, beta, film, dout):
    B, T, Cc = c.shape
    dev = c.device
    dc, dgp, dbp = torch.empty_like(c), _empty((B, Cc), dev), _empty((B, Cc), dev)
    dfilm = _empty((B, 2 * Cc), dev) if film is not None else None
    L.check(L.lib().vt_gn_mish_bwd(L.ptr(c), L.ptr(gamma), L.ptr(beta), L.ptr(film), L.ptr(dout), L.ptr(dc), L.ptr(dgp), L.ptr(dbp), L.ptr(dfilm),
                                   B, T, Cc, 8, 1e-5, _sp(dev)), "vt_gn_mish_bwd")
    return dc, colsum(dgp), colsum(dbp), dfilm


def linear_bwd(x, w, dy):
    """y = x w^T + b: -> dx, dw, db."""
    return gemm(dy, transpose(w)), gemm(transpose(dy), transpose(x)), colsum(dy)


# ---------------------------------------------------------------------------------------------- the U-Net
RB_ORDER = ["down_modules.0.0", "down_modules.0.1", "down_modules.1.0", "down_modules.1.1", "down_modules.2.0", "down_modules.2.1",
            "mid_modules.0", "mid_modules.1", "up_modules.0.0", "up_modules.0.1", "up_modules.1.0", "up_modules.1.1"]
RB_DIMS = [(10, 256), (256, 256), (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (1024, 512), (512, 512),
           (1024, 256), (256, 256)]
CIN0 = 16          # the 10 action channels padded to a multiple of 16 (GEMM k alignment); padded weight columns stay zero


def _pack_conv(w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    cout, cin, k = w.shape
    cp = cin_pad or cin
    o = torch.zeros(cout, k, cp, dtype=F32)
    o[:, :, :cin] = w.permute(0, 2, 1)
    return o.reshape(cout, k * cp)


def _unpack_conv(wp: torch.Tensor, cin: int, k: int) -> torch.Tensor:
    cout = wp.shape[0]
    return wp.reshape(cout, k, -1)[:, :, :cin].permute(0, 2, 1).contiguous()


def _pack_convT(w: torch.Tensor) -> torch.Tensor:           # [Cin][Cout][4] -> wc [Cout][4*Cin], wc[co][tp][ci] = w[ci][co][3 - tp]
    cin, cout, k = w.shape
    return w.flip(2).permute(1, 2, 0).contiguous().reshape(cout, k * cin)


def _unpack_convT(wc: torch.Tensor, cin: int) -> torch.Tensor:
    cout = wc.shape[0]
    return wc.reshape(cout, 4, cin).permute(2, 0, 1).flip(2).contiguous()


class TrainUNet:
    """One DiffusionConditionalUnet1D (input_dim 10, cond 256, dsed 256, down_dims [256, 512, 512], k 5, 8 groups) with packed fp32
    parameters `self.p`, a forward that records what the backward needs, and the backward filling `self.g` (same keys as `self.p`)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device):
        self.device = torch.device(device)
        self.p: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.g: Dict[str, torch.Tensor] = {}
        self.load_state_dict(sd)

    # ---- reference layout <-> packed layout
    def load_state_dict(self, sd):
        g = lambda k: sd[k].detach().to("cpu", F32)
        P = OrderedDict()
        P["t1.w"], P["t1.b"] = g("diffusion_step_encoder.1.weight"), g("diffusion_step_encoder.1.bias")
        P["t2.w"], P["t2.b"] = g("diffusion_step_encoder.3.weight"), g("diffusion_step_encoder.3.bias")
        P["film.w"] = torch.cat([g(f"{rb}.cond_encoder.1.weight") for rb in RB_ORDER], dim=0)
        P["film.b"] = torch.cat([g(f"{rb}.cond_encoder.1.bias") for rb in RB_ORDER], dim=0)
        for i, rb in enumerate(RB_ORDER):
            cin, cout = RB_DIMS[i]
            cp = CIN0 if cin == 10 else cin
            for j in (0, 1):
                P[f"{rb}.c{j}.w"] = _pack_conv(g(f"{rb}.blocks.{j}.block.0.weight"), cp if j == 0 else None)
                P[f"{rb}.c{j}.b"] = g(f"{rb}.blocks.{j}.block.0.bias")
                P[f"{rb}.n{j}.w"], P[f"{rb}.n{j}.b"] = g(f"{rb}.blocks.{j}.block.1.weight"), g(f"{rb}.blocks.{j}.block.1.bias")
            if cin != cout:
                P[f"{rb}.r.w"], P[f"{rb}.r.b"] = _pack_conv(g(f"{rb}.residual_conv.weight"), cp), g(f"{rb}.residual_conv.bias")
        for i in (0, 1):
            P[f"down{i}.w"], P[f"down{i}.b"] = _pack_conv(g(f"down_modules.{i}.2.conv.weight")), g(f"down_modules.{i}.2.conv.bias")
            P[f"up{i}.w"], P[f"up{i}.b"] = _pack_convT(g(f"up_modules.{i}.2.conv.weight")), g(f"up_modules.{i}.2.conv.bias")
        P["fc.w"], P["fc.b"] = _pack_conv(g("final_conv.0.block.0.weight")), g("final_conv.0.block.0.bias")
        P["fn.w"], P["fn.b"] = g("final_conv.0.block.1.weight"), g("final_conv.0.block.1.bias")
        fo = torch.zeros(CIN0, 256, dtype=F32)                               # the 10 output channels padded to 16 rows (zero rows stay zero)
        fo[:10] = _pack_conv(g("final_conv.1.weight"))
        fob = torch.zeros(CIN0, dtype=F32)
        fob[:10] = g("final_conv.1.bias")
        P["fo.w"], P["fo.b"] = fo, fob
        self.p = OrderedDict((k, v.contiguous().to(self.device)) for k, v in P.items())

    def _unpack(self, T: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
        """Packed tensors (parameters or gradients) -> the reference's keys / layouts (CPU)."""
        c = {k: v.detach().cpu() for k, v in T.items()}
        out = OrderedDict()
        out["diffusion_step_encoder.1.weight"], out["diffusion_step_encoder.1.bias"] = c["t1.w"], c["t1.b"]
        out["diffusion_step_encoder.3.weight"], out["diffusion_step_encoder.3.bias"] = c["t2.w"], c["t2.b"]
        off = 0
        for i, rb in enumerate(RB_ORDER):
            cin, cout = RB_DIMS[i]
            for j in (0, 1):
                out[f"{rb}.blocks.{j}.block.0.weight"] = _unpack_conv(c[f"{rb}.c{j}.w"], cin if j == 0 else cout, 5)
                out[f"{rb}.blocks.{j}.block.0.bias"] = c[f"{rb}.c{j}.b"]
                out[f"{rb}.blocks.{j}.block.1.weight"], out[f"{rb}.blocks.{j}.block.1.bias"] = c[f"{rb}.n{j}.w"], c[f"{rb}.n{j}.b"]
            out[f"{rb}.cond_encoder.1.weight"] = c["film.w"][off:off + 2 * cout].clone()
            out[f"{rb}.cond_encoder.1.bias"] = c["film.b"][off:off + 2 * cout].clone()
            off += 2 * cout
            if cin != cout:
                out[f"{rb}.residual_conv.weight"], out[f"{rb}.residual_conv.bias"] = _unpack_conv(c[f"{rb}.r.w"], cin, 1), c[f"{rb}.r.b"]
        for i, ch in ((0, 256), (1, 512)):
            out[f"down_modules.{i}.2.conv.weight"], out[f"down_modules.{i}.2.conv.bias"] = _unpack_conv(c[f"down{i}.w"], ch, 3), c[f"down{i}.b"]
        for i, ch in ((0, 512), (1, 256)):
            out[f"up_modules.{i}.2.conv.weight"], out[f"up_modules.{i}.2.conv.bias"] = _unpack_convT(c[f"up{i}.w"], ch), c[f"up{i}.b"]
        out["final_conv.0.block.0.weight"], out["final_conv.0.block.0.bias"] = _unpack_conv(c["fc.w"], 256, 5), c["fc.b"]
        out["final_conv.0.block.1.weight"], out["final_conv.0.block.1.bias"] = c["fn.w"], c["fn.b"]
        out["final_conv.1.weight"], out["final_conv.1.bias"] = _unpack_conv(c["fo.w"][:10].contiguous(), 256, 1), c["fo.b"][:10].clone()
        return out

    def state_dict(self):
        return self._unpack(self.p)

    def grads(self):
        return self._unpack(self.g)

    # ---- residual block
    def _rb_fwd(self, rb, x, film):
        p = self.p
        T = x.shape[1]
        c1 = conv_fwd(x, p[f"{rb}.c0.w"], p[f"{rb}.c0.b"], 5, 1, 2, T)
        f = gn_fwd(c1, p[f"{rb}.n0.w"], p[f"{rb}.n0.b"], film=film)
        c2 = conv_fwd(f, p[f"{rb}.c1.w"], p[f"{rb}.c1.b"], 5, 1, 2, T)
        res = conv_fwd(x, p[f"{rb}.r.w"], p[f"{rb}.r.b"], 1, 1, 0, T) if f"{rb}.r.w" in p else x
        out = gn_fwd(c2, p[f"{rb}.n1.w"], p[f"{rb}.n1.b"], residual=res)
        return out, (x, c1, f, c2, film)

    def _rb_bwd(self, rb, saved, dout):
        p, g = self.p, self.g
        x, c1, f, c2, film = saved
        dc2, g[f"{rb}.n1.w"], g[f"{rb}.n1.b"], _ = gn_bwd(c2, p[f"{rb}.n1.w"], p[f"{rb}.n1.b"], None, dout)
        df, g[f"{rb}.c1.w"], g[f"{rb}.c1.b"] = conv_bwd(f, p[f"{rb}.c1.w"], dc2, 5, 1, 2)
        dc1, g[f"{rb}.n0.w"], g[f"{rb}.n0.b"], dfilm = gn_bwd(c1, p[f"{rb}.n0.w"], p[f"{rb}.n0.b"], film, df)
        dx, g[f"{rb}.c0.w"], g[f"{rb}.c0.b"] = conv_bwd(x, p[f"{rb}.c0.w"], dc1, 5, 1, 2)
        if f"{rb}.r.w" in p:
            dxr, g[f"{rb}.r.w"], g[f"{rb}.r.b"] = conv_bwd(x, p[f"{rb}.r.w"], dout, 1, 1, 0)
            add_(dx, dxr)
        else:
            add_(dx, dout.contiguous())
        return dx, dfilm

    # ---- forward / backward of the whole net
    def forward(self, sample: torch.Tensor, t: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """sample [B,T,10], t [B], cond [B,256] (fp32, device) -> [B,T,10]; records the tape for `backward`."""
        p, dev = self.p, self.device
        B, T, D = sample.shape
        if T % 4 or B % 4:
            raise ValueError("TrainUNet: T and B must be multiples of 4 (fp32 GEMM k alignment of the weight-gradient products, whose reduction "
                             "runs over B or B*T/4 rows; SITrainer pads the batch with zero-weight samples)")
        pe = _empty((B, 256), dev)
        L.check(L.lib().vt_posemb(L.ptr(t), L.ptr(pe), B, 256, _sp(dev)), "vt_posemb")
        h1 = gemm(pe, p["t1.w"], p["t1.b"])
        h1m = mish(h1)
        temb = gemm(h1m, p["t2.w"], p["t2.b"])
        gfeat = _empty((B, 512), dev)
        copy_cols(temb, 0, gfeat, 0, 256)
        copy_cols(cond, 0, gfeat, 256, 256)
        gm = mish(gfeat)
        film_all = gemm(gm, p["film.w"], p["film.b"])                       # [B, 10752]
        films, off = [], 0
        for (cin, cout) in RB_DIMS:
            fl = _empty((B, 2 * cout), dev)
            copy_cols(film_all, off, fl, 0, 2 * cout)
            films.append(fl)
            off += 2 * cout
        x = torch.zeros(B, T, CIN0, dtype=F32, device=dev)
        copy_cols(sample.reshape(B * T, D), 0, x.reshape(B * T, CIN0), 0, D)
        tape = {"pe": pe, "h1": h1, "h1m": h1m, "gfeat": gfeat, "gm": gm, "rb": [None] * 12, "B": B, "T": T, "D": D}
        rb = RB_ORDER

        def run(i, x):
            out, tape["rb"][i] = self._rb_fwd(rb[i], x, films[i])
            return out
        x = run(1, run(0, x))
        tape["d0_in"] = x
        x = conv_fwd(x, p["down0.w"], p["down0.b"], 3, 2, 1, T // 2)
        x = run(3, run(2, x))
        h1s = x
        tape["d1_in"] = x
        x = conv_fwd(x, p["down1.w"], p["down1.b"], 3, 2, 1, T // 4)
        x = run(5, run(4, x))
        h2 = x
        x = run(7, run(6, x))
        xc = _empty((B, T // 4, 1024), dev)
        copy_cols(x.reshape(-1, 512), 0, xc.reshape(-1, 1024), 0, 512)
        copy_cols(h2.reshape(-1, 512), 0, xc.reshape(-1, 1024), 512, 512)
        x = run(9, run(8, xc))
        x, tape["u0_xz"] = convT_fwd(x, p["up0.w"], p["up0.b"])
        xc = _empty((B, T // 2, 1024), dev)
        copy_cols(x.reshape(-1, 512), 0, xc.reshape(-1, 1024), 0, 512)
        copy_cols(h1s.reshape(-1, 512), 0, xc.reshape(-1, 1024), 512, 512)
        x = run(11, run(10, xc))
        x, tape["u1_xz"] = convT_fwd(x, p["up1.w"], p["up1.b"])
        tape["fc_in"] = x
        cf = conv_fwd(x, p["fc.w"], p["fc.b"], 5, 1, 2, T)
        y = gn_fwd(cf, p["fn.w"], p["fn.b"])
        tape["cf"], tape["y"] = cf, y
        out16 = conv_fwd(y, p["fo.w"], p["fo.b"], 1, 1, 0, T)                  # [B, T, 16]: channels 10..15 are the zero padding
        out = _empty((B, T, D), dev)
        copy_cols(out16.reshape(B * T, CIN0), 0, out.reshape(B * T, D), 0, D)
        self._tape = tape
        return out

    def backward(self, dout: torch.Tensor) -> torch.Tensor:
        """dout [B,T,10] = d loss / d output -> fills self.g, returns d loss / d cond [B,256]."""
        p, g, tp, dev = self.p, self.g, self._tape, self.device
        B, T = tp["B"], tp["T"]
        dfilm_all = _empty((B, p["film.w"].shape[0]), dev)
        offs, o = [], 0
        for (_, cout) in RB_DIMS:
            offs.append(o)
            o += 2 * cout

        def back(i, d):
            dx, dfl = self._rb_bwd(RB_ORDER[i], tp["rb"][i], d)
            copy_cols(dfl, 0, dfilm_all, offs[i], dfl.shape[1])
            return dx
        d16 = torch.zeros(B, T, CIN0, dtype=F32, device=dev)
        copy_cols(dout.contiguous().reshape(B * T, tp["D"]), 0, d16.reshape(B * T, CIN0), 0, tp["D"])
        dy, g["fo.w"], g["fo.b"] = conv_bwd(tp["y"], p["fo.w"], d16, 1, 1, 0)
        dcf, g["fn.w"], g["fn.b"], _ = gn_bwd(tp["cf"], p["fn.w"], p["fn.b"], None, dy)
        d, g["fc.w"], g["fc.b"] = conv_bwd(tp["fc_in"], p["fc.w"], dcf, 5, 1, 2)
        d, g["up1.w"], g["up1.b"] = convT_bwd(tp["u1_xz"], p["up1.w"], d, 256)
        dxc = back(10, back(11, d))                                             # [B, T/2, 1024]
        d = _empty((B, T // 2, 512), dev)
        dh1 = _empty((B, T // 2, 512), dev)
        copy_cols(dxc.reshape(-1, 1024), 0, d.reshape(-1, 512), 0, 512)
        copy_cols(dxc.reshape(-1, 1024), 512, dh1.reshape(-1, 512), 0, 512)
        d, g["up0.w"], g["up0.b"] = convT_bwd(tp["u0_xz"], p["up0.w"], d, 512)
        dxc = back(8, back(9, d))                                               # [B, T/4, 1024]
        d = _empty((B, T // 4, 512), dev)
        dh2 = _empty((B, T // 4, 512), dev)
        copy_cols(dxc.reshape(-1, 1024), 0, d.reshape(-1, 512), 0, 512)
        copy_cols(dxc.reshape(-1, 1024), 512, dh2.reshape(-1, 512), 0, 512)
        d = back(6, back(7, d))
        add_(d, dh2)                                                            # h2 feeds the mid blocks AND the first up block
        d = back(4, back(5, d))
        d, g["down1.w"], g["down1.b"] = conv_bwd(tp["d1_in"], p["down1.w"], d, 3, 2, 1)
        add_(d, dh1)
        d = back(2, back(3, d))
        d, g["down0.w"], g["down0.b"] = conv_bwd(tp["d0_in"], p["down0.w"], d, 3, 2, 1)
        back(0, back(1, d))                                                     # (h0 is never used by the up path: no extra term)
        # FiLM Linears (concatenated), the Mish in front of them, the timestep MLP
        dgm, g["film.w"], g["film.b"] = linear_bwd(tp["gm"], p["film.w"], dfilm_all)
        dg = mish(tp["gfeat"], dgm)
        dtemb, dcond = _empty((B, 256), dev), _empty((B, 256), dev)
        copy_cols(dg, 0, dtemb, 0, 256)
        copy_cols(dg, 256, dcond, 0, 256)
        dh1m, g["t2.w"], g["t2.b"] = linear_bwd(tp["h1m"], p["t2.w"], dtemb)
        dh1 = mish(tp["h1"], dh1m)
        _, g["t1.w"], g["t1.b"] = linear_bwd(tp["pe"], p["t1.w"], dh1)
        self._tape = None
        return dcond


# ---------------------------------------------------------------------------------------------- the observation MLP
class TrainMLP:
    """Linear (-GELU-Linear)* : the reference's nn.Sequential MLPs with keys '0.weight', '0.bias', '2.*'[, '4.*'] (state_encoder
    bridge_controller.py:42-48; obs_encoder / force_encoder lstm_step_controller.py:44-60).  The first layer's K is padded to 16."""

    def __init__(self, sd: Dict[str, torch.Tensor], device):
        self.device = torch.device(device)
        self.idx = sorted(int(k.split(".")[0]) for k in sd if k.endswith(".weight"))
        first = f"{self.idx[0]}.weight"
        self.kin = sd[first].shape[1]
        self.kpad = (self.kin + 15) // 16 * 16
        w0 = torch.zeros(sd[first].shape[0], self.kpad, dtype=F32)
        w0[:, :self.kin] = sd[first].detach().to("cpu", F32)
        g = lambda k: sd[k].detach().to("cpu", F32).contiguous().to(self.device)
        self.p = OrderedDict()
        for i in self.idx:
            self.p[f"{i}.weight"] = w0.to(self.device) if i == self.idx[0] else g(f"{i}.weight")
            self.p[f"{i}.bias"] = g(f"{i}.bias")
        self.g: Dict[str, torch.Tensor] = {}
        self.first = first

    def _unpack(self, T):
        out = OrderedDict((k, v.detach().cpu()) for k, v in T.items())
        out[self.first] = out[self.first][:, :self.kin].contiguous()
        return out

    def state_dict(self):
        return self._unpack(self.p)

    def grads(self):
        return self._unpack(self.g)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        h = torch.zeros(B, self.kpad, dtype=F32, device=self.device)
        copy_cols(x, 0, h, 0, self.kin)
        tape = []
        for n, i in enumerate(self.idx):
            a = gemm(h, self.p[f"{i}.weight"], self.p[f"{i}.bias"])
            tape.append((h, a))
            h = gelu(a) if n + 1 < len(self.idx) else a
        self._tape = tape
        return h

    def backward(self, dout: torch.Tensor) -> torch.Tensor:
        """-> the gradient with respect to the (unpadded) input."""
        d = dout
        for n in range(len(self.idx) - 1, -1, -1):
            i = self.idx[n]
            h, a = self._tape[n]
            if n + 1 < len(self.idx):
                d = gelu(a, d)
            d, self.g[f"{i}.weight"], self.g[f"{i}.bias"] = linear_bwd(h, self.p[f"{i}.weight"], d)
        self._tape = None
        return d


# ---------------------------------------------------------------------------------------------- the training step
_GAMMA = {"2^0.5*t(t-1)": 0, "(2t(t-1))^0.5": 1, "(1-t)^2(2t)^0.5": 2}
# interpolant_type -> code of vt_si_qsample_ex (every string bridge_model.py:103-147 accepts)
_INTERPOLANT = {"linear": 0, "power3": 1, "power4": 2, "reverse_power3": 3, "reverse_power4": 4, "gaussian_encode_decode": 5, "reverse_linear": 6}


class _Optimizer:
    """AdamW (+ EMA shadows) over the tensors a trainer yields from `_all_params()`, eager or inside a captured graph."""
    ema_decay = 0.0
    shadow: Dict[str, torch.Tensor] = {}

    def _ema_decay(self, step: int) -> float:
        return min(self.ema_decay, (1 + step) / (10 + step))                                 # torch_ema's warm-up

    def _shadow_source(self, name: str) -> torch.Tensor:
        raise NotImplementedError

    def optimizer_step(self, hyper: Optional[torch.Tensor] = None):
        """AdamW on every trained tensor, then the EMA update of the net parameters (bridge_train.py:331-334).  `hyper` (device,
        [lr, 1-b1^t, sqrt(1-b2^t), 1-ema_decay_t]) selects the kernels that read the step-dependent scalars from memory (graph replay)."""
        if hyper is None:
            self.step_count += 1
            self.ema_updates = getattr(self, "ema_updates", 0) + 1
        lib, dev = L.lib(), self.device
        for name, p, g in self._all_params():
            if g is None:
                raise RuntimeError(f"no gradient for {name}: call get_loss first")
            if name not in self._m:
                self._m[name], self._v[name] = torch.zeros_like(p), torch.zeros_like(p)
            if hyper is None:
                L.check(lib.vt_adamw(L.ptr(p), L.ptr(g.contiguous()), L.ptr(self._m[name]), L.ptr(self._v[name]), p.numel(), self.lr, self.betas[0],
                                     self.betas[1], self.eps, self.wd, self.step_count, _sp(dev)), "vt_adamw")
        if hyper is not None:       # graph path: every tensor's AdamW (+ EMA) in one launch over a device table of pointers
            rows, chunk0 = [], 0
            for name, p, g in self._all_params():
                assert g.is_contiguous() and p.is_contiguous()
                sh = self.shadow.get(name)
                rows.append([p.data_ptr(), g.data_ptr(), self._m[name].data_ptr(), self._v[name].data_ptr(), 0 if sh is None else sh.data_ptr(),
                             p.numel(), chunk0])
                chunk0 += (p.numel() + 4095) // 4096
            self._mt_host.copy_(torch.tensor(rows, dtype=torch.int64))          # pinned, allocated by _capture_prep() before the capture began
            self._mt_dev.copy_(self._mt_host, non_blocking=True)
            L.check(lib.vt_adamw_ema_multi(L.ptr(self._mt_dev), len(rows), chunk0, L.ptr(hyper), self.betas[0], self.betas[1], self.eps, self.wd,
                                           _sp(dev)), "vt_adamw_ema_multi")
            return
        for name, sh in self.shadow.items():
            L.check(lib.vt_ema_update(L.ptr(sh), L.ptr(self._shadow_source(name)), sh.numel(), self._ema_decay(self.ema_updates), _sp(dev)), "vt_ema_update")

    # ---- the whole step as one hipGraph (the eager step is a dependent chain of hundreds to thousands of launches issued from Python)
    def _capture_prep(self, **static: torch.Tensor) -> None:
        dev = self.device
        self._st = static
        self._hyper = torch.zeros(4, dtype=F32, device=dev)
        self._hyper_ring = [torch.zeros(4, dtype=F32).pin_memory() for _ in range(8)]
        self._hyper_ev = [None] * 8
        for name, p, _ in self._all_params():
            if name not in self._m:
                self._m[name], self._v[name] = torch.zeros_like(p), torch.zeros_like(p)
        ntens = sum(1 for _ in self._all_params())
        self._mt_host = torch.zeros(ntens, 7, dtype=torch.int64).pin_memory()
        self._mt_dev = torch.zeros(ntens, 7, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()

    def _replay(self, **inputs) -> None:
        if getattr(self, "_graph", None) is None:
            raise RuntimeError("call capture(...) first (a captured step is invalidated when parameters / EMA shadows are reloaded)")
        for k, v in inputs.items():
            v = torch.as_tensor(v)
            # a pinned host tensor stays the caller's: an asynchronous copy would still be reading it when the caller refills it for the
            # next step (replays are enqueued back to back), so that one case is copied synchronously
            self._st[k].copy_(v.reshape(self._st[k].shape), non_blocking=not (v.device.type == "cpu" and v.is_pinned()))
        self.step_count += 1
        self.ema_updates = getattr(self, "ema_updates", 0) + 1
        # the step scalars travel through a RING of pinned slots, each guarded by the event recorded behind its last H2D copy: with one
        # slot, the copy queued for step n would still be waiting behind step n-1's graph when the host writes step n+1's values
        slot = self.step_count % len(self._hyper_ring)
        ev = self._hyper_ev[slot]
        if ev is not None:
            ev.synchronize()
        host = self._hyper_ring[slot]
        L.check(L.lib().vt_train_hyper(self.lr, self.betas[0], self.betas[1], self.step_count, self._ema_decay(self.ema_updates),
                                       L.ptr(host)), "vt_train_hyper")
        self._hyper.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._hyper_ev[slot] = ev
        self._graph.replay()


class SITrainer(_Optimizer):
    """get_loss + backward + AdamW + EMA for `StochasticInterpolants` (+ the observation MLP), one step per `train_step` call.

    net_sd: the reference's `InterpolantsConditionalUnet1D` state dict (keys 'v_net.*', 's_net.*', 'b_net.*'); mlp_sd: state_encoder's.
    Hyper-parameters as bridge_train.py:49-58 (AdamW) and bridge_model.py:433 (EMA decay 0.75, torch_ema warm-up)."""

    def __init__(self, net_sd, mlp_sd=None, *, gamma_type="2^0.5*t(t-1)", interpolant_type="linear", beta_max=0.03, lr=1e-4, weight_decay=1e-6,
                 betas=(0.9, 0.999), eps=1e-8, ema_decay=0.75, device="cuda"):
        if interpolant_type not in _INTERPOLANT:                   # the reference raises NotImplementedError for an unknown string too (bridge_model.py:145)
            raise NotImplementedError(interpolant_type)
        self.interpolant_type = _INTERPOLANT[interpolant_type]
        if gamma_type not in _GAMMA:
            raise NotImplementedError(gamma_type)
        self.device = L.require_gpu(device)
        self.gamma_type, self.d, self.t_min = _GAMMA[gamma_type], beta_max, 0.001
        self.nets = OrderedDict((n, TrainUNet({k[len(n) + 1:]: v for k, v in net_sd.items() if k.startswith(n + ".")}, self.device))
                                for n in ("b_net", "v_net", "s_net"))
        self.mlp = TrainMLP(mlp_sd, self.device) if mlp_sd is not None else None
        self.lr, self.wd, self.betas, self.eps, self.ema_decay = lr, weight_decay, betas, eps, ema_decay
        self.step_count = 0
        self.ema_updates = 0
        self._m: Dict[str, torch.Tensor] = {}
        self._v: Dict[str, torch.Tensor] = {}
        self.shadow = {f"{n}.{k}": v.clone() for n, u in self.nets.items() for k, v in u.p.items()}     # EMA covers net.parameters()

    def _all_params(self):
        for n, u in self.nets.items():
            for k in u.p:
                yield f"{n}.{k}", u.p[k], u.g.get(k)
        if self.mlp is not None:
            for k in self.mlp.p:
                yield f"state_encoder.{k}", self.mlp.p[k], self.mlp.g.get(k)

    def get_loss(self, obs: torch.Tensor, vla_n: torch.Tensor, expert_n: torch.Tensor, t: torch.Tensor, z: torch.Tensor, *, backward: bool = True,
                 sync: bool = True):
        """obs: `obs_cond` [B,256], or the observation MLP's input [B, 2*Dv+13] when the trainer owns the MLP; vla_n / expert_n
        [B,T,10] normalised source / target; t [B] in [0,1] (the reference's torch.rand draw); z [B,T,10] N(0,1) (scaled by beta_max
        here, as `interpolant` does) -> (loss, {'v_loss','s_loss','b_loss'}) as python floats; gradients are left in the nets."""
        dev = self.device
        f = lambda a: a.to(dev, F32).contiguous()
        obs, x0, x1, t, z = f(obs), f(vla_n), f(expert_n), f(t), f(z) * self.d
        Breal, T, D = x0.shape
        B = Breal
        while B % 4:                             # a ragged last batch: pad with samples that get no loss weight (their d loss / d out is 0,
            B += 1                               # and every layer is per-sample, so they contribute nothing to any gradient)
        if B != Breal:
            grow = lambda a, fill=0.0: torch.cat([a, torch.full((B - Breal,) + tuple(a.shape[1:]), fill, dtype=F32, device=dev)])
            obs, x0, x1, z, t = grow(obs), grow(x0), grow(x1), grow(z), grow(t, 0.5)
        cond = self.mlp.forward(obs) if self.mlp is not None else obs
        xt, tv, ts, tb = (torch.empty_like(x0) for _ in range(4))
        tc = _empty((B,), dev)
        L.check(L.lib().vt_si_qsample_ex(L.ptr(x0), L.ptr(x1), L.ptr(z), L.ptr(t), L.ptr(xt), L.ptr(tv), L.ptr(ts), L.ptr(tb), L.ptr(tc), B, T * D,
                                         self.gamma_type, self.t_min, self.interpolant_type, _sp(dev)), "vt_si_qsample_ex")
        losses, dcond = {}, None
        for name, tgt in (("v_net", tv), ("s_net", ts), ("b_net", tb)):
            out = self.nets[name].forward(xt, tc, cond)
            dout, loss = (torch.empty_like(out) if B == Breal else torch.zeros_like(out)), _empty((1,), dev)
            L.check(L.lib().vt_si_loss(L.ptr(out), L.ptr(tgt), L.ptr(dout), L.ptr(loss), Breal, T * D, _sp(dev)), "vt_si_loss")
            losses[name[0] + "_loss"] = loss
            if backward:
                dc = self.nets[name].backward(dout)
                dcond = dc if dcond is None else add_(dcond, dc)
        if backward and self.mlp is not None:
            self.mlp.backward(dcond)
        self.last_dcond = None if dcond is None else dcond[:Breal]
        if not sync:            # device tensors, no host read-back: what a captured (hipGraph) step returns
            return None, losses
        vals = {k: float(v.item()) for k, v in losses.items()}
        return vals["v_loss"] + vals["s_loss"] + vals["b_loss"], vals

    def _shadow_source(self, name: str) -> torch.Tensor:
        n, k = name.split(".", 1)
        return self.nets[n].p[k]

    def capture(self, batch: int, horizon: int = 16, dim: int = 10) -> None:
        """Record get_loss + backward + AdamW + EMA for a fixed batch shape into a hipGraph (torch.cuda.graph: stream capture of the
        C-ABI launches on torch's current stream).  `replay()` copies the inputs into the static buffers, writes the 16 bytes of
        step-dependent scalars and launches the graph; results are identical to the eager step's (same kernels, same order)."""
        z = lambda *s: torch.zeros(*s, dtype=F32, device=self.device)
        obs_dim = self.mlp.kin if self.mlp is not None else 256
        self._capture_prep(obs=z(batch, obs_dim), x0=z(batch, horizon, dim), x1=z(batch, horizon, dim), t=z(batch), z=z(batch, horizon, dim))
        with torch.cuda.graph(self._graph):
            _, self._graph_losses = self.get_loss(self._st["obs"], self._st["x0"], self._st["x1"], self._st["t"], self._st["z"], sync=False)
            self.optimizer_step(hyper=self._hyper)

    def replay(self, obs, vla_n, expert_n, t, z):
        """One captured training step -> {'v_loss','s_loss','b_loss'} device tensors (read them after a synchronize)."""
        self._replay(obs=obs, x0=vla_n, x1=expert_n, t=t, z=z)
        return self._graph_losses

    def train_step(self, obs, vla_n, expert_n, t, z, *, sync: bool = True):
        loss, info = self.get_loss(obs, vla_n, expert_n, t, z, sync=sync)
        self.optimizer_step()
        return loss, info

    def net_state_dict(self):
        return OrderedDict((f"{n}.{k}", v) for n, u in self.nets.items() for k, v in u.state_dict().items())

    def ema_state_dict(self):
        out = OrderedDict()
        for n, u in self.nets.items():
            for k, v in u._unpack({k: self.shadow[f"{n}.{k}"] for k in u.p}).items():
                out[f"{n}.{k}"] = v
        return out

    def load_ema(self, ema_sd, num_updates: int) -> None:
        """Adopt EMA shadow tensors given in the net's state-dict layout (a checkpoint's `ema.shadow_params` zipped with the net's keys)."""
        for n in self.nets:
            tmp = TrainUNet({k[len(n) + 1:]: v for k, v in ema_sd.items() if k.startswith(n + ".")}, self.device)
            for k, v in tmp.p.items():
                self.shadow[f"{n}.{k}"] = v
        # torch_ema's warm-up counter only: the AdamW step t (bias corrections) stays with the optimizer state, which a checkpoint of the
        # reference does not carry (fresh moments <-> t restarts at 0, bridge_train.py saves no optimizer.pt)
        self.ema_updates = int(num_updates)
        self._graph = None          # a captured step holds the old shadow tensors' addresses: capture again before the next replay

    def net_grads(self):
        return OrderedDict((f"{n}.{k}", v) for n, u in self.nets.items() for k, v in u.grads().items())


# ---------------------------------------------------------------------------------------------- the LSTM residual head
def _cosine_lr(base: float, step: int, t_max: int = 100000) -> float:
    """optim.lr_scheduler.CosineAnnealingLR(T_max=100000, eta_min=lr/10) (bridge_train.py:59-61, lstm_train.py:31-33), closed form."""
    import math
    eta = base / 10
    return eta + (base - eta) * (1 + math.cos(math.pi * step / t_max)) / 2


class LstmTrainer(_Optimizer):
    """forward + get_loss + BPTT + AdamW for `TactileLSTMController` (lstm_step_controller.py:176-211, 321-337; lstm_train.py:26-33,
    129-133).  `mods` = the checkpoint's 'modules' dict: obs_encoder / force_encoder (nn.Sequential MLPs), lstm (torch.nn.LSTM keys
    weight_ih_l{k}, weight_hh_l{k}, bias_ih_l{k}, bias_hh_l{k}), output_head (0 Linear, 1 LayerNorm, 4 Linear).

    Dropout (nn.LSTM inter-layer p=0.1, head p=`dropout`) is applied through explicit masks: `masks=None` is eval-mode arithmetic (what
    the parity goldens pin, the reference in `.eval()` under autograd); `masks="draw"` draws Bernoulli keep-masks on the device each step as
    training mode does; a dict {'lstm': [B,T,H] , 'head': [B,T,H]} of 0 / 1/(1-p) tensors injects them (tests)."""

    def __init__(self, mods, *, lr=1e-4, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, lstm_dropout=0.1, head_dropout=0.1, device="cuda"):
        self.device = dev = L.require_gpu(device)
        g = lambda t: t.detach().to("cpu", F32).contiguous().to(dev)
        self.obs = TrainMLP(mods["obs_encoder"], dev) if mods.get("obs_encoder") is not None else None
        self.force = TrainMLP(mods["force_encoder"], dev)
        ls = mods["lstm"]
        self.nl = len([k for k in ls if k.startswith("weight_ih_l")])
        self.H = ls["weight_hh_l0"].shape[1]
        self.lstm: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.kin, self.kpad = [], []
        for l in range(self.nl):
            w = ls[f"weight_ih_l{l}"].detach().to("cpu", F32)
            kin = w.shape[1]
            kp = (kin + 15) // 16 * 16
            wp = torch.zeros(4 * self.H, kp, dtype=F32)
            wp[:, :kin] = w
            self.kin.append(kin), self.kpad.append(kp)
            self.lstm[f"weight_ih_l{l}"] = wp.to(dev)
            for k in (f"weight_hh_l{l}", f"bias_ih_l{l}", f"bias_hh_l{l}"):
                self.lstm[k] = g(ls[k])
        hd = mods["output_head"]
        self.head = OrderedDict((k, g(hd[k])) for k in ("0.weight", "0.bias", "1.weight", "1.bias", "4.weight", "4.bias"))
        self.gl: Dict[str, torch.Tensor] = {}
        self.gh: Dict[str, torch.Tensor] = {}
        self.lr, self.base_lr, self.wd, self.betas, self.eps = lr, lr, weight_decay, betas, eps
        self.p_lstm, self.p_head = lstm_dropout, head_dropout
        self.step_count = 0
        self.ema_updates = 0
        self._m: Dict[str, torch.Tensor] = {}
        self._v: Dict[str, torch.Tensor] = {}

    # ---- one layer over the whole sequence
    def _layer_fwd(self, l, X, B, T):
        H, dev, lib, sp = self.H, self.device, L.lib(), _sp(self.device)
        bias = self.lstm[f"bias_ih_l{l}"].clone()                               # b_ih + b_hh folded into the all-tick input projection
        add_(bias, self.lstm[f"bias_hh_l{l}"])
        gx = gemm(X, self.lstm[f"weight_ih_l{l}"], bias)                  # [B*T, 4H], every tick at once
        act, cseq = _empty((B * T, 4 * H), dev), _empty((B * T, H), dev)
        hseq, hprev, hcur = _empty((B * T, H), dev), _empty((B * T, H), dev), _empty((B, H), dev)
        gh = None
        for t in range(T):
            if t > 0:
                gh = ops.gemm(hcur, self.lstm[f"weight_hh_l{l}"])
            L.check(lib.vt_lstm_cell_fwd(L.ptr(gx), L.ptr(gh), L.ptr(act), L.ptr(cseq), L.ptr(hseq), L.ptr(hprev), L.ptr(hcur), B, T, H, t, sp),
                    "vt_lstm_cell_fwd")
        return hseq, (X, act, cseq, hprev)

    def _layer_bwd(self, l, saved, dhseq, B, T):
        X, act, cseq, hprev = saved
        H, dev, lib, sp = self.H, self.device, L.lib(), _sp(self.device)
        whh_t = transpose(self.lstm[f"weight_hh_l{l}"])                         # [H, 4H]: dh_{t-1} = dgates_t W_hh
        dgates, dgcur, dc = _empty((B * T, 4 * H), dev), _empty((B, 4 * H), dev), _empty((B, H), dev)
        dh_rec = None
        for t in range(T - 1, -1, -1):
            L.check(lib.vt_lstm_cell_bwd(L.ptr(dhseq), L.ptr(dh_rec), L.ptr(act), L.ptr(cseq), L.ptr(dc), L.ptr(dgates), L.ptr(dgcur), B, T, H, t, sp),
                    "vt_lstm_cell_bwd")
            if t > 0:
                dh_rec = ops.gemm(dgcur, whh_t)
        dX, self.gl[f"weight_ih_l{l}"], db = linear_bwd(X, self.lstm[f"weight_ih_l{l}"], dgates)
        self.gl[f"weight_hh_l{l}"] = gemm(transpose(dgates), transpose(hprev))
        self.gl[f"bias_ih_l{l}"], self.gl[f"bias_hh_l{l}"] = db, db
        return dX

    def _mask(self, masks, key, shape, p):
        if masks is None or p <= 0:
            return None
        if isinstance(masks, str):
            if masks != "draw":
                raise ValueError("masks must be None, 'draw' or a dict of tensors")
            return (torch.rand(shape, device=self.device) >= p).to(F32) / (1.0 - p)
        m = masks.get(key)
        return None if m is None else m.to(self.device, F32).contiguous().reshape(shape)

    def get_loss(self, obs, vla_n, forces, expert_n, *, masks=None, backward: bool = True, sync: bool = True):
        """obs: obs_cond [B,H], or the obs_encoder's input [B, 2*Dv+state] when the trainer owns that MLP; vla_n / expert_n [B,T,D]
        normalised; forces [B,T,F] -> (loss, pred [B,T,D]); gradients are left in the trainer."""
        dev, lib, sp, H = self.device, L.lib(), _sp(self.device), self.H
        f = lambda a: torch.as_tensor(a).to(dev, F32).contiguous()
        obs, vla, forces, expert = f(obs), f(vla_n), f(forces), f(expert_n)
        Breal, T, D = vla.shape
        B = (Breal + 3) // 4 * 4                 # ragged last batch: zero-weight padding samples (k alignment of the B-row weight-gradient products)
        if B != Breal:
            grow = lambda a: torch.cat([a, torch.zeros((B - Breal,) + tuple(a.shape[1:]), dtype=F32, device=dev)])
            obs, vla, forces, expert = grow(obs), grow(vla), grow(forces), grow(expert)
        M = B * T
        cond = self.obs.forward(obs) if self.obs is not None else obs
        ef = self.force.forward(forces.reshape(M, -1))                           # [M, H/2]
        X = torch.zeros(M, self.kpad[0], dtype=F32, device=dev)                  # [encoded force | vla action | 0]
        copy_cols(ef, 0, X, 0, ef.shape[1])
        copy_cols(vla.reshape(M, D), 0, X, ef.shape[1], D)
        saved, lmasks = [], []
        for l in range(self.nl):
            hseq, sv = self._layer_fwd(l, X, B, T)
            saved.append(sv)
            if l + 1 < self.nl:
                m = self._mask(masks, "lstm" if self.nl == 2 else f"lstm{l}", (M, H), self.p_lstm)
                lmasks.append(m)
                X = hseq
                if m is not None:
                    X = hseq.clone()
                    L.check(lib.vt_mul_(L.ptr(X), L.ptr(m), X.numel(), sp), "vt_mul_")
        comb = _empty((M, 2 * H), dev)
        copy_cols(hseq, 0, comb, 0, H)
        L.check(lib.vt_bcast_mid(L.ptr(cond), L.ptr(comb), 2 * H, H, B, T, H, sp), "vt_bcast_mid")
        a1 = gemm(comb, self.head["0.weight"], self.head["0.bias"])
        n1 = ops.rownorm(a1, self.head["1.weight"], self.head["1.bias"], 1e-5)
        g1_ = gelu(n1)
        hm = self._mask(masks, "head", (M, H), self.p_head)
        if hm is not None:
            L.check(lib.vt_mul_(L.ptr(g1_), L.ptr(hm), g1_.numel(), sp), "vt_mul_")
        delta = ops.gemm(g1_, self.head["4.weight"], self.head["4.bias"])
        pred, ddelta, loss = torch.zeros(M, D, dtype=F32, device=dev), torch.zeros(M, D, dtype=F32, device=dev), _empty((1,), dev)
        L.check(lib.vt_mse_residual(L.ptr(vla), L.ptr(delta), L.ptr(expert), L.ptr(pred), L.ptr(ddelta), L.ptr(loss), Breal * T * D, sp), "vt_mse_residual")
        if backward:
            Dp = (D + 15) // 16 * 16                                              # GEMM k alignment: the D = 10 outputs padded with zeros
            ddp, w4p = torch.zeros(M, Dp, dtype=F32, device=dev), torch.zeros(Dp, H, dtype=F32, device=dev)
            copy_cols(ddelta, 0, ddp, 0, D)
            copy_cols(self.head["4.weight"], 0, w4p, 0, H)
            dg1 = ops.gemm(ddp, transpose(w4p))
            self.gh["4.weight"], self.gh["4.bias"] = gemm(transpose(ddelta), transpose(g1_)), colsum(ddelta)
            if hm is not None:
                L.check(lib.vt_mul_(L.ptr(dg1), L.ptr(hm), dg1.numel(), sp), "vt_mul_")
            dn1 = gelu(n1, dg1)
            da1, dyxh = torch.empty_like(a1), torch.empty_like(a1)
            L.check(lib.vt_ln_bwd(L.ptr(a1), L.ptr(self.head["1.weight"]), L.ptr(dn1), L.ptr(da1), L.ptr(dyxh), M, H, 1e-5, sp), "vt_ln_bwd")
            self.gh["1.weight"], self.gh["1.bias"] = colsum(dyxh), colsum(dn1)
            dcomb, self.gh["0.weight"], self.gh["0.bias"] = linear_bwd(comb, self.head["0.weight"], da1)
            dcond = _empty((B, H), dev)
            L.check(lib.vt_sum_mid(L.ptr(dcomb), 2 * H, H, L.ptr(dcond), B, T, H, sp), "vt_sum_mid")
            dh = _empty((M, H), dev)
            copy_cols(dcomb, 0, dh, 0, H)
            for l in range(self.nl - 1, -1, -1):
                dX = self._layer_bwd(l, saved[l], dh, B, T)
                if l > 0:
                    dh = dX                                                       # kpad == H for the upper layers
                    if lmasks[l - 1] is not None:
                        L.check(lib.vt_mul_(L.ptr(dh), L.ptr(lmasks[l - 1]), dh.numel(), sp), "vt_mul_")
            def_ = _empty((M, ef.shape[1]), dev)
            copy_cols(dX, 0, def_, 0, ef.shape[1])
            self.force.backward(def_)
            if self.obs is not None:
                self.obs.backward(dcond)
            self.last_dcond = dcond[:Breal]
        return (float(loss.item()) if sync else loss), pred.reshape(B, T, D)[:Breal]

    # ---- parameters in the reference's layout
    def _all(self):
        if self.obs is not None:
            for k in self.obs.p:
                yield f"obs_encoder.{k}", self.obs.p[k], self.obs.g.get(k)
        for k in self.force.p:
            yield f"force_encoder.{k}", self.force.p[k], self.force.g.get(k)
        for k in self.lstm:
            yield f"lstm.{k}", self.lstm[k], self.gl.get(k)
        for k in self.head:
            yield f"output_head.{k}", self.head[k], self.gh.get(k)

    def _unpack_lstm(self, T):
        out = OrderedDict()
        for k, v in T.items():
            v = v.detach().cpu()
            if k.startswith("weight_ih_l"):
                v = v[:, :self.kin[int(k[len("weight_ih_l"):])]].contiguous()
            out[k] = v
        return out

    def modules_state_dict(self):
        """{'obs_encoder', 'force_encoder', 'lstm', 'output_head'} as tactile_controller.pt stores them (lstm_step_controller.py:351-363)."""
        out = {"force_encoder": self.force.state_dict(), "lstm": self._unpack_lstm(self.lstm),
               "output_head": OrderedDict((k, v.detach().cpu()) for k, v in self.head.items())}
        if self.obs is not None:
            out["obs_encoder"] = self.obs.state_dict()
        return out

    def modules_grads(self):
        out = {"force_encoder": self.force.grads(), "lstm": self._unpack_lstm(self.gl), "output_head": OrderedDict((k, v.detach().cpu()) for k, v in self.gh.items())}
        if self.obs is not None:
            out["obs_encoder"] = self.obs.grads()
        return out

    def _all_params(self):
        return self._all()

    def capture(self, batch: int, horizon: int = 16, dim: int = 10, masks="draw") -> None:
        """One training step (get_loss with device-drawn dropout masks, BPTT, AdamW) as a hipGraph for a fixed batch shape."""
        z = lambda *s: torch.zeros(*s, dtype=F32, device=self.device)
        obs_dim = self.obs.kin if self.obs is not None else self.H
        self._capture_prep(obs=z(batch, obs_dim), vla=z(batch, horizon, dim), forces=z(batch, horizon, self.force.kin), expert=z(batch, horizon, dim))
        with torch.cuda.graph(self._graph):
            self._graph_loss, self._graph_pred = self.get_loss(self._st["obs"], self._st["vla"], self._st["forces"], self._st["expert"], masks=masks,
                                                               sync=False)
            self.optimizer_step(hyper=self._hyper)

    def replay(self, obs, vla_n, forces, expert_n, *, schedule: bool = True):
        """-> the loss as a 1-element device tensor (read it after a synchronize)."""
        if schedule:
            self.lr = _cosine_lr(self.base_lr, self.step_count)
        self._replay(obs=obs, vla=vla_n, forces=forces, expert=expert_n)
        return self._graph_loss

    def train_step(self, obs, vla_n, forces, expert_n, *, masks="draw", schedule: bool = True):
        if schedule:
            self.lr = _cosine_lr(self.base_lr, self.step_count)
        loss, _ = self.get_loss(obs, vla_n, forces, expert_n, masks=masks)
        self.optimizer_step()
        return loss
