"""RDT runner — mirror of the reference's VLA/models/rdt_runner.py:14-253 (`RDTRunner`, inference surface).

Same keyword-only constructor, `config` dict layout (`config['rdt']`, `config['lang_adaptor']`, ..., `config['noise_scheduler']`),
sub-module names and state-dict keys (`model.*`, `lang_adaptor.*`, `img_adaptor.*`, `state_adaptor.*`), projector types
(`linear`, `mlpNx_gelu`; ValueError otherwise), and `predict_action` / `conditional_sample` / `adapt_conditions`
signatures.  `from_pretrained(path)` reads a local HF-style directory (config.json + model.safetensors / pytorch_model.bin)
like the hub mixin (models/hub_mixin.py:26-75) — there is no network here.  Training (`compute_loss`, the DDPM scheduler)
is out of scope.

predict_action is ONE call into the HIP driver (vt_rdt_sample): adaptors, per-chunk cross-attention K/V caches,
num_inference_timesteps x depth blocks and the DPM-Solver++ updates.  The N(0,1) start of the reverse process is drawn
with torch.randn on the device (as the reference does) unless `x_init=` is given.
"""
from __future__ import annotations

import json
import os
import re
from collections import OrderedDict
from typing import Optional

import torch

from vlatouch import synth
from vlatouch.module import ParamModule
from vlatouch.engine import AutoRange
from vlatouch.rdt_engine import RdtEngine, adaptor_depth
from models.rdt.model import RDT


class _Adaptor(ParamModule):
    """nn.Linear ('weight', 'bias') or nn.Sequential of Linear / GELU(tanh) ('0.weight', '2.weight', ...)."""

    def __init__(self, projector_type, in_features, out_features, seed, materialize=True):
        self.projector_type = projector_type
        n = adaptor_depth(projector_type)
        shapes = OrderedDict()
        if n == 1:
            shapes["weight"] = (out_features, in_features)
            shapes["bias"] = (out_features,)
        else:
            for j in range(n):
                shapes[f"{2 * j}.weight"] = (out_features, in_features if j == 0 else out_features)
                shapes[f"{2 * j}.bias"] = (out_features,)
        super().__init__(shapes, device="cpu", seed=seed, materialize=materialize)


RMS_MODES = ("meansq", "var")
_warned_default_rms = False


def resolve_rms_mode(explicit: Optional[str], rdt_cfg: dict) -> str:
    """Which arithmetic timm's `RmsNorm` (models/rdt/blocks.py:22) stands for — third-party, NOT pinned by the reference:
      "meansq": x * rsqrt(mean(x^2) + eps) * w        timm >= 1.0.9 (and every textbook RMSNorm)
      "var"   : x * rsqrt(var_unbiased(x) + eps) * w  timm <= 1.0.8 — incl. timm==1.0.3, which upstream RDT-1B pins and
                                                       whose released checkpoints were therefore trained with
    Order: the `rms_mode=` constructor argument, `config['rdt']['rms_norm']`, the VLATOUCH_TIMM_RMSNORM environment variable,
    else "meansq" with a one-time warning (a silent default would hide a mismatch with the deployed timm)."""
    global _warned_default_rms
    mode = explicit or rdt_cfg.get('rms_norm') or os.environ.get("VLATOUCH_TIMM_RMSNORM")
    if mode is None:
        if not _warned_default_rms:
            import warnings
            warnings.warn("RDTRunner: RmsNorm arithmetic not specified (rms_mode= / config['rdt']['rms_norm']); using 'meansq' "
                          "(timm >= 1.0.9).  Checkpoints trained under upstream RDT-1B's pinned timm==1.0.3 need 'var'.", stacklevel=3)
            _warned_default_rms = True
        mode = "meansq"
    if mode not in RMS_MODES:
        raise ValueError(f"rms_mode must be one of {RMS_MODES}, got {mode!r}")
    return mode


class RDTRunner:
    def __init__(self, *, action_dim, pred_horizon, config, lang_token_dim, img_token_dim, state_token_dim, max_lang_cond_len,
                 img_cond_len, lang_pos_embed_config=None, img_pos_embed_config=None, dtype=torch.bfloat16, device="cuda",
                 rms_mode: Optional[str] = None, init_weights: bool = True, solver_state: Optional[str] = None, compute_dtype=None):
        hidden_size = config['rdt']['hidden_size']
        # compute_dtype (extension): the engine's 16-bit activation / MFMA operand type for a bf16 model — torch.float16 (default: the bf16 weights convert
        # exactly, same width and MFMA rate, 3 more mantissa bits: |chunk - fp32 reference| 1e-2 -> 1.2e-3 at RDT-1B) or torch.bfloat16
        # (the reference's own execution dtype, rounding after every op).  config['rdt']['compute_dtype'] / VLATOUCH_RDT_COMPUTE = "f16" | "bf16".
        # "auto" (default since round 6): start in fp16 under the engine's RANGE GUARD (vlatouch.engine.RangeGuard / AutoRange; include/vlatouch.h,
        # vt_rdt_set_range_flag): weights that do not fit fp16 -> bf16 at load; a clamped hand-off operand or a non-finite x0 prediction on the first call
        # (checked synchronously) or any later call (checked without blocking, one call of lag) -> a RuntimeWarning, the engine is rebuilt in bf16 — the
        # reference's own execution dtype, rdt_runner.py:47-60,160, model.py:124 — and the call is repeated.  "f16" / "bf16" pin the type (the guard still
        # records: `runner.engine().overflowed()`).
        cd = compute_dtype or config.get('rdt', {}).get('compute_dtype') or os.environ.get("VLATOUCH_RDT_COMPUTE", "auto")
        self._range = None
        if isinstance(cd, str):
            if cd not in ("auto", "f16", "fp16", "float16", "bf16", "bfloat16"):
                raise ValueError(f"compute_dtype must be 'auto', 'f16' or 'bf16', got {cd!r}")
            if cd == "auto" and dtype == torch.bfloat16:
                self._range = AutoRange("RDTRunner")
            cd = torch.bfloat16 if cd in ("bf16", "bfloat16") else torch.float16
        self.compute_dtype = cd if dtype == torch.bfloat16 else dtype       # fp32 (and true fp16) models compute in their own dtype
        # precision of the sampler's state between network evaluations in the 16-bit mode (extension; RdtEngine): "fp32" (default) or the reference's "bf16"
        self.solver_state = solver_state or config.get('rdt', {}).get('solver_state') or os.environ.get("VLATOUCH_RDT_SOLVER_STATE", "fp32")
        self.config = config
        self.dtype = dtype
        self.device = device
        self.rms_mode = resolve_rms_mode(rms_mode, config.get('rdt', {}))
        self._init_weights = init_weights     # False: shapes only, weights arrive via load_state_dict(..., assign=True)
        self.model = RDT(output_dim=action_dim, horizon=pred_horizon, hidden_size=hidden_size, depth=config['rdt']['depth'],
                         num_heads=config['rdt']['num_heads'], max_lang_cond_len=max_lang_cond_len, img_cond_len=img_cond_len,
                         lang_pos_embed_config=lang_pos_embed_config, img_pos_embed_config=img_pos_embed_config, dtype=dtype,
                         rms_mode=self.rms_mode, init_weights=init_weights)
        self.lang_adaptor = self.build_condition_adapter(config['lang_adaptor'], in_features=lang_token_dim, out_features=hidden_size)
        self.img_adaptor = self.build_condition_adapter(config['img_adaptor'], in_features=img_token_dim, out_features=hidden_size)
        self.state_adaptor = self.build_condition_adapter(config['state_adaptor'], in_features=state_token_dim * 2, out_features=hidden_size)
        ns = config['noise_scheduler']
        self.num_train_timesteps = ns['num_train_timesteps']
        self.num_inference_timesteps = ns['num_inference_timesteps']
        self.prediction_type = ns['prediction_type']
        self.beta_schedule = ns['beta_schedule']
        self.pred_horizon = pred_horizon
        self.action_dim = action_dim
        self.lang_token_dim, self.img_token_dim, self.state_token_dim = lang_token_dim, img_token_dim, state_token_dim
        self.max_lang_cond_len, self.img_cond_len = max_lang_cond_len, img_cond_len
        self._engine: Optional[RdtEngine] = None
        self._engine_key = None

    # ---- module plumbing
    _PARTS = ("model", "lang_adaptor", "img_adaptor", "state_adaptor")

    def build_condition_adapter(self, projector_type, in_features, out_features):
        if projector_type != 'linear' and not re.match(r'^mlp(\d+)x_gelu$', projector_type):
            raise ValueError(f'Unknown projector type: {projector_type}')
        return _Adaptor(projector_type, in_features, out_features, seed=len(projector_type) + in_features,
                        materialize=getattr(self, "_init_weights", True))

    def state_dict(self):
        sd = OrderedDict()
        for part in self._PARTS:
            for k, v in getattr(self, part).state_dict().items():
                sd[f"{part}.{k}"] = v
        return sd

    def load_state_dict(self, sd, strict=True, assign=False):
        for part in self._PARTS:
            sub = {k[len(part) + 1:]: v for k, v in sd.items() if k.startswith(part + ".")}
            getattr(self, part).load_state_dict(sub, strict=strict, assign=assign)
        extra = [k for k in sd if k.split(".")[0] not in self._PARTS]
        if strict and extra:
            raise RuntimeError(f"unexpected keys in state_dict: {extra[:5]}")
        return self

    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            self.device = device
        return self

    def eval(self):
        return self

    def parameters(self):
        for part in self._PARTS:
            yield from getattr(self, part).parameters()

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        """Local directory with config.json (the constructor kwargs) and model.safetensors or pytorch_model.bin."""
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        cfg.update(kwargs)
        dt = cfg.pop("dtype", "bfloat16")
        if isinstance(dt, str):
            dt = getattr(torch, dt.replace("torch.", ""))
        obj = cls(dtype=dt, **cfg)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        obj.load_state_dict(sd)
        return obj

    def engine(self) -> RdtEngine:
        key = tuple(getattr(self, p).version for p in self._PARTS) + (str(self.device),)
        if self._engine is None or self._engine_key != key:
            self._engine = RdtEngine(
                self.state_dict(), hidden=self.model.hidden_size, depth=self.model.depth, heads=self.model.num_heads,
                horizon=self.pred_horizon, action_dim=self.action_dim, lang_token_dim=self.lang_token_dim, img_token_dim=self.img_token_dim,
                state_token_dim=self.state_token_dim, max_lang_cond_len=self.max_lang_cond_len, img_cond_len=self.img_cond_len,
                lang_adaptor=self.config['lang_adaptor'], img_adaptor=self.config['img_adaptor'], state_adaptor=self.config['state_adaptor'],
                dtype=self.compute_dtype, io_dtype=self.dtype, rms_mode=self.rms_mode, solver_state=self.solver_state, device=self.device)
            self._engine_key = key
            rg = self._range
            if rg is not None and not rg.fell_back and self.compute_dtype == torch.float16 and not self._engine.fits_fp16:
                # static side of the guard: a weight of the bf16 checkpoint does not fit IEEE fp16 (it converted to inf), or a q / k norm gain is so large that a
                # normed q / k element could (8 |gain| > 65504)
                import warnings
                warnings.warn(f"RDTRunner: max |weight| = {self._engine.weight_absmax:g} / max q-k norm gain = {self._engine.headnorm_gain_absmax:g} does not fit IEEE fp16; "
                              "computing in bf16", RuntimeWarning, stacklevel=3)
                rg.fell_back = True
                self._to_bf16()
        return self._engine

    def _to_bf16(self):
        """The range guard's fallback: drop the fp16 engine (its 2.4 GB of converted weights first) and rebuild in the reference's bf16."""
        self._engine = None
        self.compute_dtype = torch.bfloat16
        return self.engine()

    def _guarded(self, call):
        """Run `call(engine)`; under compute_dtype="auto" consult the engine's range guard afterwards and, if it fired, repeat the call in bf16."""
        eng = self.engine()
        out = call(eng)
        rg = self._range
        if rg is not None and not rg.fell_back:
            bits = rg.after(eng, eng.dtype == torch.float16)
            if bits:
                rg.fall_back(bits)
                del eng
                out = call(self._to_bf16())
        return out

    def overflowed(self, clear: bool = False) -> int:
        """Bits of the engine's sticky range-guard word (0 = every value stayed inside the 16-bit compute type; vlatouch._lib.RANGE_NAMES); synchronises."""
        return self.engine().overflowed(clear=clear)

    # ---- inference
    def adapt_conditions(self, lang_tokens, img_tokens, state_tokens):
        """(B, L, lang_dim), (B, img_len, img_dim), (B, n, 2*state_dim) -> tokens of width hidden_size (rdt_runner.py:108-120).
        Runs the three adaptor GEMM chains on the GPU through the primitive C ABI."""
        from vlatouch import ops, _lib as L
        eng = self.engine()
        out = []
        for name, x in (("lang_adaptor", lang_tokens), ("img_adaptor", img_tokens), ("state_adaptor", state_tokens)):
            sd = getattr(self, name).state_dict()
            h = x.to(eng.device, self.dtype).reshape(-1, x.shape[-1]).contiguous()
            keys = ["weight"] if "weight" in sd else sorted((k for k in sd if k.endswith(".weight")), key=lambda k: int(k.split(".")[0]))
            for j, kw in enumerate(keys):
                w = sd[kw].to(eng.device, self.dtype).contiguous()
                b = sd[kw.replace("weight", "bias")].to(eng.device, torch.float32).contiguous()
                h = ops.gemm(h, w, b, act=L.ACT_GELU_TANH if j < len(keys) - 1 else L.ACT_NONE)
            out.append(h.reshape(*x.shape[:-1], -1))
        return tuple(out)

    def conditional_sample(self, lang_cond, lang_attn_mask, img_cond, state_traj, action_mask, ctrl_freqs, x_init=None):
        """Adapted conditions -> denoised (masked) action chunk (B, horizon, action_dim)  (rdt_runner.py:122-165)."""
        eng = self.engine()
        B = state_traj.shape[0]
        if x_init is None:
            x_init = self._draw_start(eng, B)
        with torch.no_grad():
            return self._guarded(lambda e: e.sample(lang_cond, lang_attn_mask, img_cond, state_traj.reshape(B, -1), action_mask, ctrl_freqs, x_init,
                                                    num_inference_steps=self.num_inference_timesteps, num_train_timesteps=self.num_train_timesteps,
                                                    beta_schedule=self.beta_schedule, prediction_type=self.prediction_type, adapted=True))

    def _draw_start(self, eng, B):
        """The reference's `torch.randn(size=(B, horizon, action_dim), dtype=dtype)` start (rdt_runner.py:136): torch's generator, so
        `torch.manual_seed` governs it as in the reference.  A caller that wants no torch kernel in its step passes `x_init=` drawn by
        vlatouch.ops.DeviceRng (fp32 storage, values on the dtype's grid: no cast kernels either)."""
        return torch.randn(B, self.pred_horizon, self.action_dim, dtype=self.dtype, device=eng.device)

    def predict_action(self, lang_tokens, lang_attn_mask, img_tokens, state_tokens, action_mask, ctrl_freqs, x_init=None, return_fp32=False):
        """lang_tokens (B, L, lang_dim), lang_attn_mask (B, L) bool, img_tokens (B, img_len, img_dim), state_tokens (B, 1, state_dim),
        action_mask (B, 1, action_dim) 0/1 float, ctrl_freqs (B,) -> (B, horizon, action_dim)  (rdt_runner.py:225-250).
        return_fp32 (not in the reference): hand back the fp32 buffer of the solver (un-rounded with solver_state="fp32", on the dtype grid with "bf16") instead of casting it to dtype."""
        eng = self.engine()
        B = lang_tokens.shape[0]
        if x_init is None:
            x_init = self._draw_start(eng, B)
        with torch.no_grad():
            return self._guarded(lambda e: e.sample(lang_tokens, lang_attn_mask, img_tokens, state_tokens, action_mask, ctrl_freqs, x_init,
                                                    num_inference_steps=self.num_inference_timesteps, num_train_timesteps=self.num_train_timesteps,
                                                    beta_schedule=self.beta_schedule, prediction_type=self.prediction_type, adapted=False,
                                                    return_fp32=return_fp32))

    def compute_loss(self, *a, **k):
        raise NotImplementedError("training is outside this build's scope (inference-only hot path)")

    def forward(self, *args, **kwargs):
        return self.compute_loss(*args, **kwargs)
