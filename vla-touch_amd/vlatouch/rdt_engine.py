"""RDT engine: packs an RDTRunner state dict (SURVEY Appendix A.5 key layout) and drives vt_rdt_forward /
vt_rdt_sample.  Weights stay resident in HBM in the execution dtype (bf16 = the reference's dtype, or fp32)."""
from __future__ import annotations

import ctypes as C
import re
from typing import List, Mapping, Optional

import numpy as np
import torch

from . import _lib as L
from . import dpm
from .engine import RangeGuard, _Workspace

SD = Mapping[str, torch.Tensor]


def adaptor_depth(kind: str) -> int:
    if kind == "linear":
        return 1
    m = re.match(r"^mlp(\d+)x_gelu$", kind)
    if not m:
        raise ValueError(f"Unknown projector type: {kind}")
    return int(m.group(1))


class RdtEngine(RangeGuard):
    def __init__(self, sd: SD, *, hidden: int, depth: int, heads: int, horizon: int, action_dim: int, lang_token_dim: int,
                 img_token_dim: int, state_token_dim: int, max_lang_cond_len: int, img_cond_len: int,
                 lang_adaptor: str = "mlp2x_gelu", img_adaptor: str = "mlp2x_gelu", state_adaptor: str = "mlp3x_gelu",
                 dtype: torch.dtype = torch.bfloat16, rms_mode: str = "meansq", solver_state: str = "fp32", io_dtype: Optional[torch.dtype] = None,
                 device="cuda"):
        """dtype = the engine's storage / MFMA operand type (fp32, bf16, or IEEE fp16).  io_dtype = the MODEL's dtype when it differs: a bf16 checkpoint
        evaluated with fp16 activations (dtype=torch.float16, io_dtype=torch.bfloat16: the bf16 weights convert exactly, same width and MFMA rate, 8x
        smaller activation rounding) still rounds its start noise / returns its result on the bf16 grid.
        solver_state (16-bit mode only): "fp32" (default) keeps the DPM-Solver++ state, the x0 predictions and the final projection in fp32 between
        network evaluations — closer to the fp32 reference; "bf16" reproduces the reference's bf16 rounding points (rdt_runner.py:137-139,160)."""
        self.device = L.require_gpu(device)
        if solver_state not in ("fp32", "bf16"):
            raise ValueError(f"solver_state must be 'fp32' or 'bf16', got {solver_state!r}")
        self.solver_state = solver_state
        self.io_dtype = io_dtype or dtype
        if hidden // heads != 64:
            raise L.VtError("RdtEngine: head_dim must be 64")
        self.dtype = dtype
        cdt = L.dt_code(dtype)
        dev = self.device
        self.cfg = dict(hidden=hidden, depth=depth, heads=heads, horizon=horizon, action_dim=action_dim)
        self.hidden, self.horizon, self.action_dim, self.img_len, self.max_lang = hidden, horizon, action_dim, img_cond_len, max_lang_cond_len
        self.state_dim, self.lang_dim, self.img_dim = state_token_dim, lang_token_dim, img_token_dim
        f32 = torch.float32
        w = lambda k: sd[k].detach().to(dev, dtype).contiguous()
        v = lambda k: sd[k].detach().to(dev, f32).contiguous()
        W: List[torch.Tensor] = []
        for e in ("t_embedder", "freq_embedder"):
            W += [w(f"model.{e}.mlp.0.weight"), v(f"model.{e}.mlp.0.bias"), w(f"model.{e}.mlp.2.weight"), v(f"model.{e}.mlp.2.bias")]
        W += [v("model.x_pos_embed")[0].contiguous(), w("model.lang_cond_pos_embed")[0].contiguous(), w("model.img_cond_pos_embed")[0].contiguous()]
        for i in range(depth):
            p = f"model.blocks.{i}"
            W += [v(f"{p}.norm1.weight"), w(f"{p}.attn.qkv.weight"), v(f"{p}.attn.qkv.bias"), v(f"{p}.attn.q_norm.weight"), v(f"{p}.attn.k_norm.weight"),
                  w(f"{p}.attn.proj.weight"), v(f"{p}.attn.proj.bias"),
                  v(f"{p}.norm2.weight"), w(f"{p}.cross_attn.q.weight"), v(f"{p}.cross_attn.q.bias"), w(f"{p}.cross_attn.kv.weight"),
                  v(f"{p}.cross_attn.kv.bias"), v(f"{p}.cross_attn.q_norm.weight"), v(f"{p}.cross_attn.k_norm.weight"),
                  w(f"{p}.cross_attn.proj.weight"), v(f"{p}.cross_attn.proj.bias"),
                  v(f"{p}.norm3.weight"), w(f"{p}.ffn.fc1.weight"), v(f"{p}.ffn.fc1.bias"), w(f"{p}.ffn.fc2.weight"), v(f"{p}.ffn.fc2.bias")]
        fl = "model.final_layer"
        W += [v(f"{fl}.norm_final.weight"), w(f"{fl}.ffn_final.fc1.weight"), v(f"{fl}.ffn_final.fc1.bias"), w(f"{fl}.ffn_final.fc2.weight"),
              v(f"{fl}.ffn_final.fc2.bias")]
        depths = []
        for name, kind in (("lang_adaptor", lang_adaptor), ("img_adaptor", img_adaptor), ("state_adaptor", state_adaptor)):
            n = adaptor_depth(kind)
            depths.append(n)
            if n == 1:
                W += [w(f"{name}.weight"), v(f"{name}.bias")]
            else:
                for j in range(n):
                    W += [w(f"{name}.{2 * j}.weight"), v(f"{name}.{2 * j}.bias")]
        self._weights = W
        d = L.RdtDesc()
        d.hidden, d.depth, d.heads, d.horizon, d.out_dim, d.state_dim = hidden, depth, heads, horizon, action_dim, state_token_dim
        d.lang_dim, d.img_dim, d.max_lang_len, d.img_len = lang_token_dim, img_token_dim, max_lang_cond_len, img_cond_len
        d.n_lang, d.n_img, d.n_state = depths
        d.cdt = d.adt = cdt
        d.rms_mode = {"meansq": L.NORM_RMS_MEANSQ, "var": L.NORM_RMS_VAR}[rms_mode]
        lib = L.lib()
        assert lib.vt_rdt_num_weights(C.byref(d)) == len(W), (lib.vt_rdt_num_weights(C.byref(d)), len(W))
        self._h = C.c_void_p()
        L.check(lib.vt_rdt_create(C.byref(d), L.ptr_array(W), len(W), C.byref(self._h)), "vt_rdt_create")
        L.check(lib.vt_rdt_set_state_precision(self._h, int(solver_state == "fp32")), "vt_rdt_set_state_precision")
        if dtype != torch.float32 and self.io_dtype != torch.float32:
            L.check(lib.vt_rdt_set_io_dtype(self._h, L.dt_code(self.io_dtype)), "vt_rdt_set_io_dtype")
        self._range_init(lib.vt_rdt_set_range_flag, "vt_rdt_set_range_flag")
        # static side of the range guard: can the weights themselves be held in this 16-bit type?  (a bf16 checkpoint converted to IEEE fp16: |w| > 65504 became inf)
        mats = [t for t in W if t.dtype == dtype and t.dtype != torch.float32]
        self.weight_absmax = float(torch.stack([t.abs().max().float() for t in mats]).max()) if mats else 0.0
        # the per-head q / k RMSNorm gains (weight order: csrc/vt_rdt.hip): a normed q / k element is at most 8 |gain| (mean-square form), stored in 16 bits
        hn = [W[11 + 21 * i + j] for i in range(depth) for j in (3, 4, 12, 13)]
        self.headnorm_gain_absmax = float(torch.stack([t.abs().max() for t in hn]).max())
        self.fits_fp16 = self.weight_absmax <= 65504.0 and 8.0 * self.headnorm_gain_absmax <= 65504.0
        self._ws = _Workspace(dev)
        # frozen weights of the denoise-loop Linears, a second time in MFMA fragment order (csrc/vt_gemm_pw.hip streams them global -> VGPR)
        self._depth, self._rms_mode = depth, rms_mode
        self._set_score_bounds()
        nb = lib.vt_rdt_packed_bytes(self._h)
        self._packed = None
        if nb:
            self._packed = torch.empty(nb, dtype=torch.uint8, device=dev)
            L.check(lib.vt_rdt_set_packed(self._h, L.ptr(self._packed), L.stream_ptr(dev)), "vt_rdt_set_packed")

    def _set_score_bounds(self):
        """|q . k| / 8 <= 8 max_i |w_q[i] w_k[i]| for the mean-square per-head RMSNorm of cross_attn.q_norm / k_norm (blocks.py:86-87, 112-113):
        q = q^ * w_q with |q^|_2 <= 8, likewise k, so |sum_i q^_i w_q[i] k^_i w_k[i]| <= max_i |w_q[i] w_k[i]| |q^| |k^| (round 6: the maximum of the PRODUCTS,
        not the product of the maxima) (+2 %: the normed q / k are stored in 16 bits): lets the cached cross-attention drop its running maximum — for bounds
        up to 40 with bf16 probabilities, up to 10 with IEEE fp16 ones (csrc/vt_attn_kvt.hip).  The variance form has no such bound -> 0 (online softmax).
        Weight order per block: see csrc/vt_rdt.hip (cq_norm at +12, ck_norm at +13)."""
        bounds = (C.c_float * self._depth)()
        if self._rms_mode == "meansq" and self.dtype in (torch.bfloat16, torch.float16):
            for i in range(self._depth):
                base = 11 + 21 * i
                wq, wk = self._weights[base + 12], self._weights[base + 13]
                bounds[i] = 8.0 * 1.02 * float((wq * wk).abs().max())
        self.score_bounds = [float(b) for b in bounds]      # 0 = unknown: that block's cross-attention keeps the online softmax
        L.check(L.lib().vt_rdt_set_score_bounds(self._h, bounds, self._depth), "vt_rdt_set_score_bounds")

    def repack(self):
        """Re-derive what was computed from `_weights` at load time after they changed in place (e.g. the multi-GPU weight broadcast):
        the fragment-packed copies and the cross-attention score bounds."""
        self._set_score_bounds()
        if self._packed is not None:
            L.check(L.lib().vt_rdt_set_packed(self._h, L.ptr(self._packed), L.stream_ptr(self.device)), "vt_rdt_set_packed")

    def update_weights(self, fn):
        """In-place change of the packed weights through `fn(self._weights)`, followed by `repack()` (the derived copies never go stale)."""
        fn(self._weights)
        self.repack()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().vt_rdt_destroy(self._h)
        except Exception:
            pass

    @staticmethod
    def _expect(name, t, shape):
        """The C driver indexes raw pointers with the packed config: a wrong shape would read out of bounds (the reference
        raises a broadcast / matmul error in the same situations), so every tensor is checked here."""
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"RdtEngine: {name} has shape {tuple(t.shape)}, expected {tuple(shape)}")

    def _ws_for(self, B, Llang):
        return self._ws.get(L.lib().vt_rdt_workspace_bytes(self._h, B, Llang))

    def forward(self, x, freq, t, lang_c, img_c, lang_mask=None) -> torch.Tensor:
        """RDT.forward: x [B, horizon+1, D], freq [B], t [B] or [1], lang_c [B,L,D], img_c [B,img_len,D] -> [B, horizon, out]."""
        dev, dt = self.device, self.dtype
        if x.dim() != 3 or lang_c.dim() != 3:
            raise ValueError(f"RdtEngine.forward: x and lang_c must be 3-D, got {tuple(x.shape)} and {tuple(lang_c.shape)}")
        B, Llang = x.shape[0], lang_c.shape[1]
        if not 1 <= Llang <= self.max_lang:
            raise ValueError(f"RdtEngine.forward: language length {Llang} outside 1..{self.max_lang}")
        self._expect("x", x, (B, self.horizon + 1, self.hidden))
        self._expect("lang_c", lang_c, (B, Llang, self.hidden))
        self._expect("img_c", img_c, (B, self.img_len, self.hidden))
        self._expect("freq", freq, (B,))
        if lang_mask is not None:
            self._expect("lang_mask", lang_mask, (B, Llang))
        if torch.as_tensor(t).numel() not in (1, B):
            raise ValueError(f"RdtEngine.forward: t must have 1 or {B} elements, got {torch.as_tensor(t).numel()}")
        x = x.to(dev, dt).contiguous()
        lang_c, img_c = lang_c.to(dev, dt).contiguous(), img_c.to(dev, dt).contiguous()
        freq = freq.to(dev, torch.float32).contiguous()
        t = torch.as_tensor(t).to(torch.float32).reshape(-1)
        scalar = t.numel() == 1
        tdev = None if scalar else t.to(dev).contiguous()
        mask = None if lang_mask is None else lang_mask.to(dev).contiguous()
        if mask is not None:
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        out = torch.empty(B, self.horizon, self.action_dim, dtype=dt, device=dev)
        L.check(L.lib().vt_rdt_forward(self._h, L.ptr(x), L.ptr(freq), L.ptr(tdev), float(t[0]) if scalar else 0.0, int(scalar), L.ptr(lang_c),
                                       L.ptr(img_c), L.ptr(mask), L.ptr(out), B, Llang, L.ptr(self._ws_for(B, Llang)), L.stream_ptr(dev)),
                "vt_rdt_forward")
        self.range_poll()
        return out

    def sample(self, lang_tokens, lang_attn_mask, img_tokens, state_tokens, action_mask, ctrl_freqs, x_init, *, num_inference_steps: int,
               num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2", prediction_type: str = "sample",
               adapted: bool = False, return_fp32: bool = False) -> torch.Tensor:
        """RDTRunner.predict_action (adapted=False: raw encoder tokens) / conditional_sample (adapted=True: tokens already
        projected to hidden size); x_init = the N(0,1) start [B, horizon, action_dim]."""
        if prediction_type not in ("sample", "epsilon"):
            raise ValueError(f"Unsupported prediction type {prediction_type}")
        dev, dt = self.device, self.dtype
        if lang_tokens.dim() != 3:
            raise ValueError(f"RdtEngine.sample: lang_tokens must be [B, L, dim], got {tuple(lang_tokens.shape)}")
        B, Llang = lang_tokens.shape[0], lang_tokens.shape[1]
        if not 1 <= Llang <= self.max_lang:
            raise ValueError(f"RdtEngine.sample: language length {Llang} outside 1..{self.max_lang}")
        if num_inference_steps < 1:
            raise ValueError("RdtEngine.sample: num_inference_steps must be >= 1")
        self._expect("lang_tokens", lang_tokens, (B, Llang, self.hidden if adapted else self.lang_dim))
        self._expect("lang_attn_mask", lang_attn_mask, (B, Llang))
        self._expect("img_tokens", img_tokens, (B, self.img_len, self.hidden if adapted else self.img_dim))
        if adapted:
            if state_tokens.numel() != B * self.hidden:
                raise ValueError(f"RdtEngine.sample: adapted state token has shape {tuple(state_tokens.shape)}, expected [{B}, 1, {self.hidden}]")
        else:
            self._expect("state_tokens", state_tokens, (B, 1, self.state_dim))
        self._expect("action_mask", action_mask, (B, 1, self.state_dim))
        self._expect("ctrl_freqs", ctrl_freqs, (B,))
        self._expect("x_init", x_init, (B, self.horizon, self.action_dim))
        ts, coef = dpm.schedule(num_train_timesteps, beta_schedule, num_inference_steps)
        ts_c = (C.c_float * len(ts))(*[float(t) for t in ts])
        coef_c = (C.c_float * coef.size)(*coef.reshape(-1).tolist())
        lang_tokens = lang_tokens.to(dev, dt).contiguous()
        img_tokens = img_tokens.to(dev, dt).contiguous()
        state_tokens = state_tokens.to(dev, dt).contiguous()
        action_mask = action_mask.to(dev, dt).contiguous()
        ctrl_freqs = ctrl_freqs.to(dev, torch.float32).contiguous()
        x_init = x_init.to(dev, torch.float32).contiguous()
        mask = lang_attn_mask.to(dev).contiguous()
        mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)        # bool is one byte of 0 / 1: no cast kernel
        out = torch.empty(B, self.horizon, self.action_dim, dtype=torch.float32, device=dev)
        L.check(L.lib().vt_rdt_sample(self._h, L.ptr(lang_tokens), L.ptr(mask), L.ptr(img_tokens), L.ptr(state_tokens), L.ptr(action_mask),
                                      L.ptr(ctrl_freqs), L.ptr(x_init), len(ts), ts_c, coef_c, int(prediction_type == "sample"), int(adapted), L.ptr(out), B, Llang,
                                      L.ptr(self._ws_for(B, Llang)), L.stream_ptr(dev)), "vt_rdt_sample")
        self.range_poll()            # range guard: asynchronous read-out of the sticky word (no synchronisation; RangeGuard)
        return out if return_fp32 else out.to(self.io_dtype)
