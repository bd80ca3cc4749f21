"""Engines: frozen-weight packing (load time) + handles into libvlatouch_hip.so (run time).

Each engine takes a plain state dict (the reference's checkpoint key layout),
re-lays the weights once for the gfx950 kernels (tap-major implicit-GEMM conv weights, fused QKV,
parity-split ConvTranspose, K padded to 16), keeps them resident in HBM and drives the C ABI.
torch is used for allocation, one-time weight re-layout and stream handles only.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib as L

SD = Mapping[str, torch.Tensor]


def _pad_to(n: int, m: int = 16) -> int:
    return (n + m - 1) // m * m


def precision_dtypes(precision: str):
    """'fp32': fp32 weights + activations (fp32 MFMA, 1e-4 parity).  'bf16': bf16 weights, fp32 accumulate."""
    if precision == "fp32":
        return L.F32, L.F32
    if precision == "bf16":
        return L.BF16, L.BF16
    if precision == "fp16":
        return L.F16, L.F16
    raise ValueError(f"precision must be 'fp32', 'bf16' or 'fp16', got {precision!r}")


class _Workspace:
    """Scratch buffer of an engine, ONE PER HIP STREAM: calls enqueued on different streams (two batches in flight, bench.py
    --streams) must not share scratch, calls on one stream are ordered and do."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[int, torch.Tensor] = {}

    def get(self, nbytes: int) -> torch.Tensor:
        key = torch.cuda.current_stream(self.device).cuda_stream
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[key] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
        return buf


class RangeGuard:
    """Mixin of the engines with a 16-bit mode: ONE uint32 in device memory that the engine's kernels OR bits into when a value left the 16-bit type's range
    (include/vlatouch.h, vt_rdt_set_range_flag / vt_dino_set_range_flag; bit names: vlatouch._lib.RANGE_NAMES).  Sticky across calls and hipGraph replays.
      overflowed()        synchronises the device and returns the bits (0 = clean); `clear=True` re-zeroes the word
      range_poll()        never blocks: returns the bits of the last asynchronous read-out that has completed and enqueues the next one on the current
                          stream (a pinned 4-byte copy + an event; skipped while the stream is being captured into a hipGraph) — one call of lag"""

    def _range_init(self, setter, what: str):
        self._range = torch.zeros(1, dtype=torch.int32, device=self.device)
        L.check(setter(self._h, L.ptr(self._range)), what)
        self._range_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._range_evt: Optional[torch.cuda.Event] = None
        self._range_seen = 0

    def overflowed(self, clear: bool = False) -> int:
        torch.cuda.synchronize(self.device)
        bits = int(self._range.item()) | self._range_seen
        if clear:
            self._range.zero_()
            self._range_seen = 0
            self._range_evt = None
        return bits

    def range_poll(self) -> int:
        if torch.cuda.is_current_stream_capturing():
            return self._range_seen
        if self._range_evt is not None and self._range_evt.query():
            self._range_seen |= int(self._range_host[0])
            self._range_evt = None
        if self._range_evt is None:
            self._range_host.copy_(self._range, non_blocking=True)
            self._range_evt = torch.cuda.Event()
            self._range_evt.record(torch.cuda.current_stream(self.device))
        return self._range_seen


class AutoRange:
    """Policy of the owners of a RangeGuard engine that runs a bf16 / fp32-trained model with IEEE fp16 storage BY DEFAULT (RDTRunner compute_dtype="auto",
    the ViT encoders' low-precision mode): the first call is checked synchronously (weight-driven overflow shows on any input), later calls through the
    engine's non-blocking read-out (one call of lag); nothing is read while a hipGraph is being captured.  `after()` returns the bits that ask for the
    fallback to bf16 storage (0 = keep going); the owner rebuilds its engine and repeats the call."""

    def __init__(self, what: str):
        self.what, self.checked, self.fell_back, self.bits = what, False, False, 0

    def after(self, eng: RangeGuard, is_f16: bool) -> int:
        if not is_f16 or torch.cuda.is_current_stream_capturing():
            return 0
        if not self.checked:
            self.checked = True
            return eng.overflowed()
        return eng.range_poll()

    def fall_back(self, bits: int) -> None:
        import warnings
        self.fell_back, self.bits = True, bits
        warnings.warn(f"{self.what}: activations left the IEEE fp16 range ({', '.join(L.range_names(bits))}); switching this model to bf16 storage "
                      "(the reference's execution dtype) for this and all later calls", RuntimeWarning, stacklevel=4)


# =========================================================================================== U-Net / SI sampler
def _conv_tapmajor(w: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """Conv1d weight [Cout, Cin, k] -> [Cout, k*cin_pad] with W'[co, tap*cin_pad + ci] = w[co, ci, tap]."""
    cout, cin, k = w.shape
    o = torch.zeros(cout, k, cin_pad, dtype=w.dtype)
    o[:, :, :cin] = w.permute(0, 2, 1)
    return o.reshape(cout, k * cin_pad)


def _convT_parity(w: torch.Tensor, taps) -> torch.Tensor:
    """ConvTranspose1d weight [Cin, Cout, 4] -> [Cout, 2*Cin] for the given pair of kernel taps."""
    return torch.cat([w[:, :, t].t() for t in taps], dim=1).contiguous()


class UNetEngine:
    """`nets` structurally identical conditional 1-D U-Nets evaluated together (v_net + s_net for the sampler).

    sds: one state dict per net with keys relative to the net (e.g. 'down_modules.0.0.blocks.0.block.0.weight').
    """

    def __init__(self, sds: Sequence[SD], *, input_dim: int = 10, global_cond_dim: int = 256, dsed: int = 256,
                 down_dims: Sequence[int] = (256, 512, 512), kernel_size: int = 5, n_groups: int = 8,
                 precision: str = "fp32", act_dtype: Optional[str] = None, device="cuda"):
        self.device = L.require_gpu(device)
        self.nets = len(sds)
        # precision: 'fp32'       exact fp32 MFMA (v_mfma_f32_16x16x4_f32), 1e-4 parity
        #            'bf16'       split-bf16: fp32 storage, a_hi*w_hi + a_lo*w_hi + a_hi*w_lo on the bf16 MFMA pipe
        #                         (the SDE multiplies the score by up to ~8/step, plain bf16 misses the 1e-2 bar)
        #            'bf16_plain' bf16 weights + activations, one bf16 MFMA per k-step (fastest, ~2.5e-2 on a_hat)
        if precision == "bf16":
            cdt, adt = L.F32X3, L.F32
        elif precision == "bf16_plain":
            cdt, adt = L.BF16, L.BF16
        else:
            cdt, adt = precision_dtypes(precision)
        if act_dtype is not None:
            adt = {"fp32": L.F32, "bf16": L.BF16}[act_dtype]
        self.cdt, self.adt = cdt, adt
        self.input_dim, self.cond_dim, self.dsed = input_dim, global_cond_dim, dsed
        self.dims = list(down_dims)
        self.k = kernel_size
        wdt = torch.bfloat16 if cdt == L.BF16 else torch.float32
        Lv = len(self.dims)
        all_dims = [input_dim] + self.dims
        rbs = []
        for l in range(Lv):
            rbs.append((f"down_modules.{l}.0", all_dims[l], self.dims[l]))
            rbs.append((f"down_modules.{l}.1", self.dims[l], self.dims[l]))
        rbs.append(("mid_modules.0", self.dims[-1], self.dims[-1]))
        rbs.append(("mid_modules.1", self.dims[-1], self.dims[-1]))
        for u in range(Lv - 1):
            din, dout = self.dims[Lv - 2 - u], self.dims[Lv - 1 - u]
            rbs.append((f"up_modules.{u}.0", 2 * dout, din))
            rbs.append((f"up_modules.{u}.1", din, din))

        # cache CPU fp32 copies once
        cpu = [{k: v.detach().float().cpu() for k, v in sd.items()} for sd in sds]

        def st(fn, dtype):
            return torch.stack([fn(sd) for sd in cpu]).to(dtype).contiguous().to(self.device)

        f32 = torch.float32
        W: List[Optional[torch.Tensor]] = []
        W.append(st(lambda sd: sd["diffusion_step_encoder.1.weight"], wdt))
        W.append(st(lambda sd: sd["diffusion_step_encoder.1.bias"], f32))
        W.append(st(lambda sd: sd["diffusion_step_encoder.3.weight"], wdt))
        W.append(st(lambda sd: sd["diffusion_step_encoder.3.bias"], f32))
        W.append(st(lambda sd: torch.cat([sd[f"{p}.cond_encoder.1.weight"] for p, _, _ in rbs], dim=0), wdt))
        W.append(st(lambda sd: torch.cat([sd[f"{p}.cond_encoder.1.bias"] for p, _, _ in rbs], dim=0), f32))
        for p, cin, cout in rbs:
            cpad = _pad_to(cin)
            W.append(st(lambda sd: _conv_tapmajor(sd[f"{p}.blocks.0.block.0.weight"], cpad), wdt))
            W.append(st(lambda sd: sd[f"{p}.blocks.0.block.0.bias"], f32))
            W.append(st(lambda sd: sd[f"{p}.blocks.0.block.1.weight"], f32))
            W.append(st(lambda sd: sd[f"{p}.blocks.0.block.1.bias"], f32))
            W.append(st(lambda sd: _conv_tapmajor(sd[f"{p}.blocks.1.block.0.weight"], cout), wdt))
            W.append(st(lambda sd: sd[f"{p}.blocks.1.block.0.bias"], f32))
            W.append(st(lambda sd: sd[f"{p}.blocks.1.block.1.weight"], f32))
            W.append(st(lambda sd: sd[f"{p}.blocks.1.block.1.bias"], f32))
            if cin != cout:
                W.append(st(lambda sd: _conv_tapmajor(sd[f"{p}.residual_conv.weight"], cpad), wdt))
                W.append(st(lambda sd: sd[f"{p}.residual_conv.bias"], f32))
            else:
                W += [None, None]
        for l in range(Lv - 1):
            W.append(st(lambda sd: _conv_tapmajor(sd[f"down_modules.{l}.2.conv.weight"], self.dims[l]), wdt))
            W.append(st(lambda sd: sd[f"down_modules.{l}.2.conv.bias"], f32))
        for u in range(Lv - 1):
            W.append(st(lambda sd: _convT_parity(sd[f"up_modules.{u}.2.conv.weight"], (1, 3)), wdt))
            W.append(st(lambda sd: _convT_parity(sd[f"up_modules.{u}.2.conv.weight"], (0, 2)), wdt))
            W.append(st(lambda sd: sd[f"up_modules.{u}.2.conv.bias"], f32))
        W.append(st(lambda sd: _conv_tapmajor(sd["final_conv.0.block.0.weight"], self.dims[0]), wdt))
        W.append(st(lambda sd: sd["final_conv.0.block.0.bias"], f32))
        W.append(st(lambda sd: sd["final_conv.0.block.1.weight"], f32))
        W.append(st(lambda sd: sd["final_conv.0.block.1.bias"], f32))
        W.append(st(lambda sd: sd["final_conv.1.weight"][:, :, 0], wdt))
        W.append(st(lambda sd: sd["final_conv.1.bias"], f32))
        self._weights = W      # keep alive: the handle stores raw device pointers

        desc = L.UnetDesc()
        desc.nets, desc.input_dim, desc.input_pad = self.nets, input_dim, _pad_to(input_dim)
        desc.cond_dim, desc.dsed, desc.n_groups, desc.ksize, desc.n_levels = global_cond_dim, dsed, n_groups, kernel_size, Lv
        for i, d in enumerate(self.dims):
            desc.dims[i] = d
        desc.cdt, desc.adt = cdt, adt
        lib = L.lib()
        n = lib.vt_unet_num_weights(C.byref(desc))
        assert n == len(W), (n, len(W))
        self._h = C.c_void_p()
        L.check(lib.vt_unet_create(C.byref(desc), L.ptr_array(W), len(W), C.byref(self._h)), "vt_unet_create")
        # fused sampler path (split-bf16 mode): pre-split hi / lo weights in MFMA fragment order, packed once on the device
        self._fused = None
        nb = lib.vt_unet_fused_bytes(self._h)
        if nb:
            self._fused = torch.empty(nb, dtype=torch.uint8, device=self.device)
            self.repack()
        self._ws = _Workspace(self.device)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().vt_unet_destroy(self._h)
        except Exception:
            pass

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        return self._ws.get(L.lib().vt_unet_workspace_bytes(self._h, B, T))

    def repack(self) -> None:
        """Rebuild the derived weight copy of the fused path from `self._weights` (after they were overwritten in place, e.g. by the
        one-time broadcast of rank 0's weights, vlatouch/dist.py)."""
        if self._fused is not None:
            L.check(L.lib().vt_unet_fused_pack(self._h, L.ptr(self._fused), L.stream_ptr(self.device)), "vt_unet_fused_pack")

    def forward(self, x: torch.Tensor, t, cond: torch.Tensor) -> torch.Tensor:
        """x [B,T,dim], t scalar or [B], cond [B,G]  ->  [nets,B,T,dim] fp32."""
        B, T, D = x.shape
        x = x.to(self.device, torch.float32).contiguous()
        cond = cond.to(self.device, torch.float32).contiguous()
        out = torch.empty(self.nets, B, T, D, dtype=torch.float32, device=self.device)
        if torch.is_tensor(t) and t.numel() > 1:
            tdev = t.to(self.device, torch.float32).contiguous()
            tp, th = L.ptr(tdev), 0.0
        else:
            tp, th = C.c_void_p(0), float(t)
        ws = self._workspace(B, T)
        L.check(L.lib().vt_unet_forward(self._h, L.ptr(x), tp, C.c_float(th), L.ptr(cond), L.ptr(out), B, T, L.ptr(ws),
                                        L.stream_ptr(self.device)), "vt_unet_forward")
        return out

    def sample(self, x0: torch.Tensor, cond: torch.Tensor, noise: Optional[torch.Tensor], n_steps: int, beta_max: float,
               record: bool = False, gamma_type: int = 0, epsilon_type: int = 0, sde_type: int = 0, backward: bool = False,
               score_weight: float = 1.0, own_x0: bool = False):
        """Velocity-score / drift-score SDE from x0 (normalised actions).  Returns xT (and the n_steps+1 states).  The integrator runs in place on
        a copy of x0 — or on x0 itself when the caller gives it away (`own_x0`: a temporary already on the device in fp32, contiguous)."""
        B, T, D = x0.shape
        if own_x0 and x0.device == self.device and x0.dtype == torch.float32 and x0.is_contiguous():
            x = x0
        else:
            x = x0.to(self.device, torch.float32).clone().contiguous()
        cond = cond.to(self.device, torch.float32).contiguous()
        if noise is not None:
            noise = noise.to(self.device, torch.float32).contiguous()
            assert tuple(noise.shape) == (n_steps, B, T, D), noise.shape
        traj = torch.empty(n_steps + 1, B, T, D, dtype=torch.float32, device=self.device) if record else None
        ws = self._workspace(B, T)
        L.check(L.lib().vt_si_sample_ex(self._h, L.ptr(x), L.ptr(cond), L.ptr(noise), n_steps, C.c_float(beta_max), gamma_type, epsilon_type,
                                        sde_type, int(bool(backward)), C.c_float(score_weight), L.ptr(traj), B, T, L.ptr(ws),
                                        L.stream_ptr(self.device)), "vt_si_sample")
        return (x, traj) if record else x


# =========================================================================================== DINOv2
class DinoEngine(RangeGuard):
    """HF Dinov2Model state dict -> CLS features (pooler_output)."""

    def __init__(self, sd: SD, *, heads: int, precision: str = "fp32", device="cuda", patch: int = 14, eps: float = 1e-6):
        self.device = L.require_gpu(device)
        cdt, adt = precision_dtypes(precision)
        self.cdt, self.adt = cdt, adt
        wdt = L.torch_dtype(cdt)
        f32 = torch.float32
        g = lambda k: sd[k].detach().float().cpu()
        pw = g("embeddings.patch_embeddings.projection.weight")
        D = pw.shape[0]
        self.hidden, self.heads, self.patch = D, heads, patch
        self.kpad = _pad_to(3 * patch * patch, 16 if cdt == L.F32 else 64)   # 16-bit: K % 64 keeps the patch GEMM on the LDS-DMA path
        layers = 0
        while f"encoder.layer.{layers}.norm1.weight" in sd:
            layers += 1
        self.layers = layers
        # dinov2-giant: gated FFN (HF Dinov2SwiGLUFFN: weights_in [2F, D] -> silu(x1) * x2 -> weights_out [D, F]); the others: fc1 / GELU / fc2
        self.swiglu = "encoder.layer.0.mlp.weights_in.weight" in sd
        fc1n, fc2n = ("mlp.weights_in", "mlp.weights_out") if self.swiglu else ("mlp.fc1", "mlp.fc2")
        dev = self.device
        pwp = torch.zeros(D, self.kpad)
        pwp[:, : 3 * patch * patch] = pw.reshape(D, -1)
        self._pos_full = g("embeddings.position_embeddings")            # [1, 1+n, D] fp32 (CPU)
        cls_pos0 = (g("embeddings.cls_token")[0, 0] + self._pos_full[0, 0]).contiguous()
        W: List[torch.Tensor] = [pwp.to(wdt).to(dev), g("embeddings.patch_embeddings.projection.bias").to(dev), cls_pos0.to(dev)]
        for i in range(layers):
            p = f"encoder.layer.{i}"
            a = f"{p}.attention.attention"
            qkv_w = torch.cat([g(f"{a}.query.weight"), g(f"{a}.key.weight"), g(f"{a}.value.weight")], dim=0)
            qkv_b = torch.cat([g(f"{a}.query.bias"), g(f"{a}.key.bias"), g(f"{a}.value.bias")], dim=0)
            W += [g(f"{p}.norm1.weight").to(dev), g(f"{p}.norm1.bias").to(dev), qkv_w.to(wdt).contiguous().to(dev), qkv_b.to(dev),
                  g(f"{p}.attention.output.dense.weight").to(wdt).to(dev), g(f"{p}.attention.output.dense.bias").to(dev),
                  g(f"{p}.layer_scale1.lambda1").to(dev),
                  g(f"{p}.norm2.weight").to(dev), g(f"{p}.norm2.bias").to(dev),
                  g(f"{p}.{fc1n}.weight").to(wdt).to(dev), g(f"{p}.{fc1n}.bias").to(dev),
                  g(f"{p}.{fc2n}.weight").to(wdt).to(dev), g(f"{p}.{fc2n}.bias").to(dev),
                  g(f"{p}.layer_scale2.lambda1").to(dev)]
        W += [g("layernorm.weight").to(dev), g("layernorm.bias").to(dev)]
        W = [w.contiguous() for w in W]
        self._weights = W
        desc = L.DinoDesc()
        desc.hidden, desc.layers, desc.heads, desc.patch, desc.kpad = D, layers, heads, patch, self.kpad
        desc.cdt, desc.adt, desc.eps = cdt, adt, eps
        if self.swiglu:
            desc.act, desc.mlp_dim = L.ACT_SWIGLU, sd[f"encoder.layer.0.{fc2n}.weight"].shape[1]
        lib = L.lib()
        assert lib.vt_dino_num_weights(C.byref(desc)) == len(W)
        self._h = C.c_void_p()
        L.check(lib.vt_dino_create(C.byref(desc), L.ptr_array(W), len(W), C.byref(self._h)), "vt_dino_create")
        # fragment-packed second copies of the fc1 weights (16-bit GELU models): the rows a 256-row tiling of the token matrix leaves over, and the CLS-only last
        # block, run fc1 on the small-M packed-weight tile (csrc/vt_gemm_pws.hip)
        nb = lib.vt_dino_packed_bytes(self._h) if os.environ.get("VLATOUCH_DINO_PACKED", "1") != "0" else 0     # =0: A/B without the packed fc1 copy
        if nb:
            self._packed = torch.empty(nb, dtype=torch.uint8, device=self.device)
            L.check(lib.vt_dino_set_packed(self._h, L.ptr(self._packed), L.stream_ptr(self.device)), "vt_dino_set_packed")
            torch.cuda.synchronize(self.device)
        self._range_init(lib.vt_dino_set_range_flag, "vt_dino_set_range_flag")
        self._ws = _Workspace(self.device)
        self._pos_cache: Dict[int, torch.Tensor] = {}
        self.last_flags: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().vt_dino_destroy(self._h)
        except Exception:
            pass

    def repack(self) -> None:
        """Rebuild the fragment-packed fc1 copy from `self._weights` (after they were overwritten in place, e.g. by the one-time broadcast of rank 0's
        weights, vlatouch/dist.py).

        INVARIANT: `self._packed` (a second, fragment-ordered copy of every fc1 weight) is DERIVED from `self._weights`; the remainder rows of the
        token matrix and the CLS-only last block read it while all other rows read `_weights`.  Any in-place write to `_weights` must be followed
        by `repack()` — use `update_weights()` for that, never write the tensors and walk away."""
        if getattr(self, "_packed", None) is not None:
            L.check(L.lib().vt_dino_set_packed(self._h, L.ptr(self._packed), L.stream_ptr(self.device)), "vt_dino_set_packed")

    def update_weights(self, fn) -> None:
        """The one sanctioned way to change weights in place: `fn(list_of_packed_device_tensors)` mutates them (checkpoint reload into the same
        tensors, a broadcast, a fine-tuning step), then every derived copy is rebuilt."""
        fn(self._weights)
        self.repack()

    def pos_patch(self, grid: int) -> torch.Tensor:
        """Position embeddings of the patch tokens for a grid x grid image: input-independent, so the bicubic
        resize of HF `interpolate_pos_encoding` (modeling_dinov2.py:57-95) is done once per resolution at load."""
        if grid not in self._pos_cache:
            pos = self._pos_full
            n = pos.shape[1] - 1
            s = int(round(math.sqrt(n)))
            pp = pos[:, 1:]
            if grid * grid != n:
                pp = pp.reshape(1, s, s, -1).permute(0, 3, 1, 2)
                pp = F.interpolate(pp, size=(grid, grid), mode="bicubic", align_corners=False)
                pp = pp.permute(0, 2, 3, 1).reshape(1, grid * grid, -1)
            self._pos_cache[grid] = pp[0].contiguous().to(self.device)
        return self._pos_cache[grid]

    def forward(self, cams: Sequence[torch.Tensor], *, nhwc: bool, pre_scale: float = 1.0, norm_mode: int = L.IMGNORM_AUTO) -> torch.Tensor:
        """cams: list of image batches (same shape/dtype, fp32 or uint8) -> [ncams, B, hidden] fp32.
        Each camera batch gets its own max()/mean() decision (visual_encoder.py:78,100)."""
        x0 = cams[0]
        is_u8 = all(c.dtype == torch.uint8 for c in cams)      # one element type per call: mixed batches are read as fp32
        cams = [c.to(self.device).contiguous() if is_u8 else c.to(self.device, torch.float32).contiguous() for c in cams]
        B = cams[0].shape[0]
        res = cams[0].shape[1] if nhwc else cams[0].shape[2]
        for c in cams:
            if c.dim() != 4 or c.shape != cams[0].shape:
                raise ValueError(f"DinoEngine.forward: every camera batch must be 4-D with one shape, got {[tuple(k.shape) for k in cams]}")
            if (c.shape[-1] if nhwc else c.shape[1]) != 3:
                raise ValueError(f"DinoEngine.forward: expected 3 colour channels, got shape {tuple(c.shape)} (nhwc={nhwc})")
            if not ((c.shape[1] == c.shape[2]) if nhwc else (c.shape[2] == c.shape[3])):
                raise ValueError(f"DinoEngine.forward: square frames only, got shape {tuple(c.shape)}")
        if res < self.patch:
            raise ValueError(f"DinoEngine.forward: frames of {res} px are smaller than one {self.patch}-px patch")
        grid = res // self.patch
        n = len(cams)
        out = torch.empty(n, B, self.hidden, dtype=torch.float32, device=self.device)
        flags = torch.empty(n, 4, dtype=torch.float32, device=self.device)
        lib = L.lib()
        ws = self._ws.get(lib.vt_dino_workspace_bytes(self._h, n * B, res))
        L.check(lib.vt_dino_forward(self._h, L.ptr_array(cams), n, int(is_u8), int(nhwc), C.c_float(pre_scale), norm_mode, B, res,
                                    L.ptr(self.pos_patch(grid)), L.ptr(out), L.ptr(flags), L.ptr(ws), L.stream_ptr(self.device)),
                "vt_dino_forward")
        self.last_flags = flags
        return out


# =========================================================================================== SigLIP image tokens
class SiglipEngine(RangeGuard):
    """HF SiglipVisionModel state dict -> last_hidden_state [B, tokens, hidden] (the RDT image tower).

    Runs on the same ViT driver as DINOv2 (vt_dino_*): no CLS token, tanh-GELU, every token through the final LayerNorm.
    The 72-wide heads are packed zero-padded to 96 (q/k/v rows, out_proj columns) so the head_dim-96 attention kernel and the
    K % 64 GEMM tiles apply; the FFN width (4304) is zero-padded to a multiple of 64 (gelu(0) = 0, zero fc2 columns)."""

    HD_PAD = 96

    def __init__(self, sd: SD, *, heads: int, precision: str = "fp32", device="cuda", patch: int = 14, eps: float = 1e-6):
        self.device = L.require_gpu(device)
        cdt, adt = precision_dtypes(precision)
        self.cdt, self.adt = cdt, adt
        wdt = L.torch_dtype(cdt)
        if any(k.startswith("vision_model.") for k in sd):
            sd = {k[len("vision_model."):]: v for k, v in sd.items() if k.startswith("vision_model.")}
        g = lambda k: sd[k].detach().float().cpu()
        pw = g("embeddings.patch_embedding.weight")
        D = pw.shape[0]
        hd = D // heads
        if hd * heads != D or hd > self.HD_PAD or D % 64:
            raise L.VtError(f"SiglipEngine: hidden {D} / heads {heads} not supported (head_dim <= 96, hidden % 64 == 0)")
        # head width on the device: 64 as is; otherwise the next multiple of 16 (16-bit modes: 72 -> 80 = two 32-deep MFMA steps + one 16-deep)
        # or 96 (fp32 mode, whose attention kernel takes 32-deep steps only); zero padding is exact
        HP = 64 if hd == 64 else ((hd + 15) // 16 * 16 if (cdt != L.F32 and hd <= 80 and hd > 64) else self.HD_PAD)
        Da = heads * HP
        self.hidden, self.heads, self.patch, self.head_dim = D, heads, patch, hd
        self.kpad = _pad_to(3 * patch * patch, 16 if cdt == L.F32 else 64)
        layers = 0
        while f"encoder.layers.{layers}.layer_norm1.weight" in sd:
            layers += 1
        self.layers = layers
        inter = g("encoder.layers.0.mlp.fc1.weight").shape[0]
        Dm = _pad_to(inter, 64)
        dev = self.device
        pwp = torch.zeros(D, self.kpad)
        pwp[:, : 3 * patch * patch] = pw.reshape(D, -1)
        self._pos = g("embeddings.position_embedding.weight").contiguous().to(dev)          # [n_pos, D] fp32
        ones = torch.ones(D)

        def pad_rows(w):          # [D, K] (rows = heads x hd) -> [Da, K]
            out = torch.zeros(heads, HP, w.shape[1])
            out[:, :hd] = w.reshape(heads, hd, w.shape[1])
            return out.reshape(Da, w.shape[1])

        def pad_vec(b):
            out = torch.zeros(heads, HP)
            out[:, :hd] = b.reshape(heads, hd)
            return out.reshape(Da)

        W: List[torch.Tensor] = [pwp.to(wdt).to(dev), g("embeddings.patch_embedding.bias").to(dev), torch.zeros(D).to(dev)]
        for i in range(layers):
            p = f"encoder.layers.{i}"
            a = f"{p}.self_attn"
            qkv_w = torch.cat([pad_rows(g(f"{a}.q_proj.weight")), pad_rows(g(f"{a}.k_proj.weight")), pad_rows(g(f"{a}.v_proj.weight"))], dim=0)
            qkv_b = torch.cat([pad_vec(g(f"{a}.q_proj.bias")), pad_vec(g(f"{a}.k_proj.bias")), pad_vec(g(f"{a}.v_proj.bias"))], dim=0)
            ow = g(f"{a}.out_proj.weight")                                                   # [D, D] columns = heads x hd
            owp = torch.zeros(D, heads, HP)
            owp[:, :, :hd] = ow.reshape(D, heads, hd)
            fc1 = torch.zeros(Dm, D)
            fc1[:inter] = g(f"{p}.mlp.fc1.weight")
            fc1b = torch.zeros(Dm)
            fc1b[:inter] = g(f"{p}.mlp.fc1.bias")
            fc2 = torch.zeros(D, Dm)
            fc2[:, :inter] = g(f"{p}.mlp.fc2.weight")
            W += [g(f"{p}.layer_norm1.weight").to(dev), g(f"{p}.layer_norm1.bias").to(dev), qkv_w.to(wdt).contiguous().to(dev), qkv_b.to(dev),
                  owp.reshape(D, Da).to(wdt).contiguous().to(dev), g(f"{a}.out_proj.bias").to(dev), ones.to(dev),
                  g(f"{p}.layer_norm2.weight").to(dev), g(f"{p}.layer_norm2.bias").to(dev),
                  fc1.to(wdt).to(dev), fc1b.to(dev), fc2.to(wdt).to(dev), g(f"{p}.mlp.fc2.bias").to(dev), ones.to(dev)]
        W += [g("post_layernorm.weight").to(dev), g("post_layernorm.bias").to(dev)]
        W = [w.contiguous() for w in W]
        self._weights = W
        desc = L.DinoDesc()
        desc.hidden, desc.layers, desc.heads, desc.patch, desc.kpad = D, layers, heads, patch, self.kpad
        desc.cdt, desc.adt, desc.eps = cdt, adt, eps
        desc.no_cls, desc.act, desc.head_dim, desc.mlp_dim, desc.out_all = 1, L.ACT_GELU_TANH, HP, Dm, 1
        desc.attn_scale = float(hd) ** -0.5
        lib = L.lib()
        assert lib.vt_dino_num_weights(C.byref(desc)) == len(W)
        self._h = C.c_void_p()
        L.check(lib.vt_dino_create(C.byref(desc), L.ptr_array(W), len(W), C.byref(self._h)), "vt_dino_create (siglip)")
        self._range_init(lib.vt_dino_set_range_flag, "vt_dino_set_range_flag")
        self._ws = _Workspace(self.device)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().vt_dino_destroy(self._h)
        except Exception:
            pass

    @property
    def num_patches(self) -> int:
        return self._pos.shape[0]

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values [B, 3, res, res] (already rescaled + normalised by the image processor) -> [B, tokens, hidden] fp32."""
        x = pixel_values.to(self.device, torch.float32).contiguous()
        B, _, res, res2 = x.shape
        grid = res // self.patch
        if res != res2 or grid * grid != self._pos.shape[0]:
            raise L.VtError(f"SiglipEngine: {res}x{res2} input does not match the {self._pos.shape[0]}-entry position table")
        out = torch.empty(B, grid * grid, self.hidden, dtype=torch.float32, device=self.device)
        flags = torch.empty(1, 4, dtype=torch.float32, device=self.device)
        lib = L.lib()
        ws = self._ws.get(lib.vt_dino_workspace_bytes(self._h, B, res))
        L.check(lib.vt_dino_forward(self._h, L.ptr_array([x]), 1, 0, 0, C.c_float(1.0), L.IMGNORM_OFF, B, res, L.ptr(self._pos), L.ptr(out),
                                    L.ptr(flags), L.ptr(ws), L.stream_ptr(self.device)), "vt_dino_forward (siglip)")
        return out


# =========================================================================================== MLP chain
class MlpEngine:
    """nn.Sequential(Linear, act, Linear, act, ..., Linear) with torch-style keys '0.weight', '2.weight', ..."""

    def __init__(self, sd: SD, *, act: int = L.ACT_GELU_ERF, precision: str = "fp32", device="cuda"):
        self.device = L.require_gpu(device)
        self.cdt, self.adt = precision_dtypes(precision)
        wdt = L.torch_dtype(self.cdt)
        idx = sorted(int(k.split(".")[0]) for k in sd if k.endswith(".weight") and sd[k].dim() == 2)
        self.W, self.b, dims = [], [], []
        for j, i in enumerate(idx):
            w = sd[f"{i}.weight"].detach().float().cpu()
            kin = _pad_to(w.shape[1])
            if j > 0:
                assert w.shape[1] == dims[-1] or kin == dims[-1]
            wp = torch.zeros(w.shape[0], kin)
            wp[:, : w.shape[1]] = w
            if j == 0:
                dims.append(kin)
                self.in_features = w.shape[1]
            assert w.shape[0] % 4 == 0 or j == len(idx) - 1
            dims.append(_pad_to(w.shape[0]) if j < len(idx) - 1 else w.shape[0])
            if j < len(idx) - 1 and dims[-1] != w.shape[0]:
                wp = torch.cat([wp, torch.zeros(dims[-1] - w.shape[0], kin)], dim=0)
            self.W.append(wp.to(wdt).contiguous().to(self.device))
            bb = sd[f"{i}.bias"].detach().float().cpu()
            if bb.numel() < dims[-1]:
                bb = torch.cat([bb, torch.zeros(dims[-1] - bb.numel())])
            self.b.append(bb.contiguous().to(self.device))
        self.dims = dims
        self.act = act
        self.out_features = dims[-1]
        self.in_pad = dims[0]
        self._dims_c = (C.c_int * len(dims))(*dims)
        self._Wp = L.ptr_array(self.W)
        self._bp = L.ptr_array(self.b)
        self._tmp = None

    def __call__(self, x: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """x: [B, in_pad] of the activation dtype (zero padded)."""
        B = x.shape[0]
        assert x.shape[1] == self.in_pad and x.dtype == L.torch_dtype(self.adt) and x.is_contiguous()
        y = torch.empty(B, self.out_features, dtype=out_dtype, device=self.device)
        mx = max(self.dims[1:-1]) if len(self.dims) > 2 else 1
        need = 2 * B * mx
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=L.torch_dtype(self.adt), device=self.device)
        L.check(L.lib().vt_mlp(L.ptr(x), x.shape[1], B, len(self.W), self._dims_c, self._Wp, self._bp, self.act, L.ptr(y),
                               L.dt_code(out_dtype), self.out_features, self.cdt, self.adt, L.ptr(self._tmp), L.stream_ptr(self.device)),
                "vt_mlp")
        return y

    def pad_input(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        xp = torch.zeros(B, self.in_pad, dtype=L.torch_dtype(self.adt), device=self.device)
        xp[:, : x.shape[1]] = x.to(self.device)
        return xp


def concat_obs(cls1, cls2, state, forces, in_pad: int, adt: int, device) -> torch.Tensor:
    """[cls_cam1 | cls_cam2 | state | forces] -> zero-padded [B, in_pad] activation matrix (bridge_controller.py:129-132)."""
    B, dv = cls1.shape
    out = torch.empty(B, in_pad, dtype=L.torch_dtype(adt), device=device)
    state = state.to(device, torch.float32).contiguous()
    fd = 0
    if forces is not None:
        forces = forces.to(device, torch.float32).contiguous()
        fd = forces.shape[1]
    L.check(L.lib().vt_concat_obs(L.ptr(cls1), L.ptr(cls2), dv, L.ptr(state), state.shape[1], L.ptr(forces), fd, L.ptr(out), adt,
                                  in_pad, B, L.stream_ptr(device)), "vt_concat_obs")
    return out


def action_normalize(x: torch.Tensor, mins: torch.Tensor, maxs: torch.Tensor, denorm: bool, padding_factor: float = 1.4) -> torch.Tensor:
    x = x.to(torch.float32).contiguous()
    dev = x.device
    mins = mins.to(dev, torch.float32).contiguous()
    maxs = maxs.to(dev, torch.float32).contiguous()
    out = torch.empty_like(x)
    L.check(L.lib().vt_action_normalize(L.ptr(x), L.ptr(out), L.ptr(mins), L.ptr(maxs), x.numel(), x.shape[-1],
                                        C.c_float(padding_factor), int(denorm), L.stream_ptr(dev)), "vt_action_normalize")
    return out


# =========================================================================================== LSTM residual head
class LstmEngine:
    def __init__(self, mods: Mapping[str, SD], *, state_dim: int = 10, hidden: int = 256, layers: int = 2, force_dim: int = 3,
                 precision: str = "fp32", device="cuda"):
        self.device = L.require_gpu(device)
        cdt = L.F32X3 if precision == "x3" else precision_dtypes(precision)[0]     # "x3": fp32 weights as bf16 hi + lo (csrc/vt_lstm.hip)
        self.cdt = cdt
        wdt = L.torch_dtype(cdt)
        dev = self.device
        H = hidden
        self.state_dim, self.hidden, self.layers = state_dim, hidden, layers
        fpad, inpad = _pad_to(force_dim), _pad_to(H // 2 + state_dim)
        if H not in (128, 256, 384):
            raise L.VtError("LstmEngine: hidden_dim must be 128, 256 or 384 (each of the persistent kernel's 8 waves owns hidden/8 = whole 16-unit tiles)")
        if state_dim > 16:
            raise L.VtError("LstmEngine: state_dim <= 16")
        g = lambda m, k: mods[m][k].detach().float().cpu()

        def pack(w: torch.Tensor, row_starts, kpad: int) -> torch.Tensor:
            """[N, K] -> MFMA fragment order [tile][k-step][lane][8]: element (lane, j) of (tile, ks) = w[row_starts[tile] + lane % 16]
            [ks*32 + (lane // 16)*8 + j]; rows / columns beyond the matrix are zero.  A wave's load instruction then reads one
            contiguous KiB (csrc/vt_lstm.hip)."""
            N, K = w.shape
            wp = torch.zeros(max(max(row_starts) + 16, N), kpad)
            wp[:N, :K] = w
            lane = torch.arange(64)
            rows = torch.tensor(row_starts)[:, None] + (lane % 16)[None, :]                      # [tiles, 64]
            cols = (torch.arange(kpad // 32) * 32)[:, None, None] + ((lane // 16) * 8)[None, :, None] + torch.arange(8)[None, None, :]   # [ks, 64, 8]
            out = wp[rows[:, None, :, None], cols[None, :, :, :]]                                # [tiles, ks, 64, 8]
            if cdt == L.F32X3:        # [tiles, ks, 64, (hi 8 | lo 8)] bf16: per lane 16 B of hi then 16 B of lo
                hi = out.to(torch.bfloat16)
                lo = (out - hi.float()).to(torch.bfloat16)
                return torch.cat([hi, lo], dim=-1).contiguous().to(dev)
            return out.contiguous().to(wdt).to(dev)

        seq_tiles = lambda n: [16 * i for i in range(n // 16)]
        UT, UW = H // 128, H // 8                   # 16-unit tiles per gate per wave, hidden units per wave
        gate_tiles = [q * H + UW * w_ + 16 * t for w_ in range(8) for q in range(4) for t in range(UT)]       # tile 4 UT w + UT q + t
        KX = (H // 2 + 16 + 31) // 32 * 32          # layer-0 input [force features H/2 | vla_n | 0 ...] in whole k-steps (H = 256: 160)
        W = [pack(g("force_encoder", "0.weight"), seq_tiles(H // 2), 32), g("force_encoder", "0.bias").to(dev),
             pack(g("force_encoder", "2.weight"), seq_tiles(H // 2), H // 2), g("force_encoder", "2.bias").to(dev)]
        for l in range(layers):
            wih, whh = g("lstm", f"weight_ih_l{l}"), g("lstm", f"weight_hh_l{l}")
            kx = KX if l == 0 else H
            wcat = torch.zeros(4 * H, kx + H)
            wcat[:, : wih.shape[1]] = wih
            wcat[:, kx:] = whh
            W += [pack(wcat, gate_tiles, kx + H), (g("lstm", f"bias_ih_l{l}") + g("lstm", f"bias_hh_l{l}")).to(dev)]
        hb2 = torch.zeros(16)
        hb2[:state_dim] = g("output_head", "4.bias")
        W += [pack(g("output_head", "0.weight"), seq_tiles(H), 2 * H), g("output_head", "0.bias").to(dev),
              g("output_head", "1.weight").to(dev), g("output_head", "1.bias").to(dev),
              pack(g("output_head", "4.weight"), [0], H), hb2.to(dev)]
        W = [w.contiguous() for w in W]
        self._weights = W
        desc = L.LstmDesc()
        desc.state_dim, desc.hidden, desc.layers, desc.force_dim, desc.force_pad, desc.in_pad, desc.cdt = \
            state_dim, H, layers, force_dim, fpad, inpad, cdt
        lib = L.lib()
        assert lib.vt_lstm_num_weights(C.byref(desc)) == len(W)
        self._h = C.c_void_p()
        L.check(lib.vt_lstm_create(C.byref(desc), L.ptr_array(W), len(W), C.byref(self._h)), "vt_lstm_create")
        self._ws = _Workspace(dev)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().vt_lstm_destroy(self._h)
        except Exception:
            pass

    def step(self, obs_cond, vla_n, force, h, c) -> torch.Tensor:
        """One tick; h, c [layers,B,H] fp32 are updated in place; returns normalised vla_n + delta."""
        dev = self.device
        B = vla_n.shape[0]
        obs_cond = obs_cond.to(dev, torch.float32).contiguous()
        vla_n = vla_n.to(dev, torch.float32).contiguous()
        force = force.to(dev, torch.float32).contiguous()
        assert h.is_contiguous() and c.is_contiguous() and h.dtype == torch.float32
        out = torch.empty(B, self.state_dim, dtype=torch.float32, device=dev)
        lib = L.lib()
        ws = self._ws.get(lib.vt_lstm_workspace_bytes(self._h, B))
        L.check(lib.vt_lstm_step(self._h, L.ptr(obs_cond), L.ptr(vla_n), L.ptr(force), L.ptr(h), L.ptr(c), L.ptr(out), B, L.ptr(ws),
                                 L.stream_ptr(dev)), "vt_lstm_step")
        return out

    def sequence(self, obs_cond, vla_n, force, h, c) -> torch.Tensor:
        """T ticks in ONE kernel launch: vla_n [B,T,S], force [B,T,F]; h, c [layers,B,H] fp32 updated in place -> [B,T,S]."""
        dev = self.device
        B, T, _ = vla_n.shape
        obs_cond = obs_cond.to(dev, torch.float32).contiguous()
        vla_n = vla_n.to(dev, torch.float32).contiguous()
        force = force.to(dev, torch.float32).contiguous()
        if tuple(force.shape[:2]) != (B, T) or obs_cond.shape != (B, self.hidden):
            raise ValueError(f"LstmEngine.sequence: obs_cond {tuple(obs_cond.shape)} / force {tuple(force.shape)} do not match vla {tuple(vla_n.shape)}")
        assert h.is_contiguous() and c.is_contiguous() and h.dtype == torch.float32 and tuple(h.shape) == (self.layers, B, self.hidden)
        out = torch.empty(B, T, self.state_dim, dtype=torch.float32, device=dev)
        L.check(L.lib().vt_lstm_sequence(self._h, L.ptr(obs_cond), L.ptr(vla_n), L.ptr(force), L.ptr(h), L.ptr(c), L.ptr(out), B, T,
                                         L.stream_ptr(dev)), "vt_lstm_sequence")
        return out
