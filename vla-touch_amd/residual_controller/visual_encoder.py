"""DINOv2 CLS-feature encoder — mirror of the reference's VLA/residual_controller/visual_encoder.py:9-106.

Same constructor and `forward(images) -> [B, hidden]` contract (numpy / BHWC / BTHWC / BCHW inputs, the
`max() > 1 -> /255` and `mean() < 0.5 -> skip ImageNet normalisation` decisions taken over the WHOLE call's
batch), but the arithmetic is the HIP DINOv2 engine (vt_dino_forward): the two data-dependent decisions are
evaluated ON DEVICE by a reduction kernel (no host sync), the patchify kernel applies them while it reads
the frames once, and the transformer runs on MFMA.

Weights: the reference calls `Dinov2Model.from_pretrained(model_name)`.  There is no hub access here, so
weights come from (in order): the `state_dict=` argument; a local directory / file `model_name` holding
`model.safetensors` or `pytorch_model.bin` (HF layout, key prefix optional); the HF cache if the snapshot is
already on disk; or, when VLATOUCH_SYNTH_WEIGHTS=1, the deterministic synthetic weights used by the
tests and bench.py.  Anything else raises FileNotFoundError (never a silent random init).
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from vlatouch import _lib as L
from vlatouch import synth
from vlatouch.engine import AutoRange, DinoEngine
from vlatouch.module import default_precision

_SIZES = {  # visual_encoder.py:31-46
    "small": dict(hidden=384, layers=12, heads=6),
    "base": dict(hidden=768, layers=12, heads=12),
    "large": dict(hidden=1024, layers=24, heads=16),
    "giant": dict(hidden=1536, layers=40, heads=24),
}


def _size_of(model_name: str) -> str:
    for s in ("small", "base", "large", "giant"):
        if s in model_name:
            return s
    return "small"


def _load_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def _find_weights(model_name: str) -> Optional[Dict[str, torch.Tensor]]:
    cands = []
    if os.path.isfile(model_name):
        cands.append(model_name)
    if os.path.isdir(model_name):
        cands += [os.path.join(model_name, f) for f in ("model.safetensors", "pytorch_model.bin")]
    hub = os.path.join(os.environ.get("HF_HOME", os.path.expanduser("~/.cache/huggingface")), "hub",
                       "models--" + model_name.replace("/", "--"), "snapshots", "*")
    for snap in sorted(glob.glob(hub)):
        cands += [os.path.join(snap, f) for f in ("model.safetensors", "pytorch_model.bin")]
    for c in cands:
        if os.path.isfile(c):
            sd = _load_file(c)
            return {k[len("dinov2."):] if k.startswith("dinov2.") else k: v for k, v in sd.items()}
    return None


class DINOv2Encoder:
    """DINOv2 vision encoder that processes images for the controller (visual_encoder.py:9)."""

    def __init__(self, model_name="facebook/dinov2-small", device="cuda", precision: Optional[str] = None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None):
        self.device = device
        self.model_name = model_name
        size = _size_of(model_name)
        cfg = _SIZES[size]
        self.patch_size = 14
        self.hidden_size = cfg["hidden"]
        self.precision = precision or default_precision()
        sd = state_dict if state_dict is not None else _find_weights(model_name)
        if sd is None:
            if os.environ.get("VLATOUCH_SYNTH_WEIGHTS") == "1":
                shapes = synth.dinov2_shapes(cfg["hidden"], cfg["layers"], swiglu=(size == "giant"))
                sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, prefix=f"dinov2-{size}.").items()}
            else:
                raise FileNotFoundError(
                    f"no DINOv2 weights for {model_name!r}: pass state_dict=, point model_name at a local HF directory, "
                    "or set VLATOUCH_SYNTH_WEIGHTS=1 for deterministic synthetic weights (no network access here)")
        self._state_dict = sd
        # the low-precision mode of the ENCODER is IEEE fp16 (same MFMA rate as bf16, 3 more mantissa bits; the residual stream
        # stays fp32 so activations are range-safe): bf16 features alone cost 7e-3 on obs_cond, most of the 1e-2 budget on a_t.
        # VLATOUCH_DINO_PRECISION=bf16 restores bf16 storage.
        # Range guard (round 6): fp16 storage is the DEFAULT of that mode, not a promise — the engine flags a saturated SwiGLU gate / a non-finite feature
        # (DinoEngine.overflowed()), and unless VLATOUCH_DINO_PRECISION pins the type this encoder then rebuilds itself with bf16 storage and repeats the call.
        eng_prec = self.precision
        self._range = None
        if eng_prec == "bf16":
            eng_prec = os.environ.get("VLATOUCH_DINO_PRECISION") or "fp16"
            if "VLATOUCH_DINO_PRECISION" not in os.environ:
                self._range = AutoRange(f"DINOv2Encoder({model_name})")
        self._heads = cfg["heads"]
        self.engine = DinoEngine(sd, heads=cfg["heads"], precision=eng_prec, device=device)
        self.model = self          # the reference exposes `.model` (eval / to / parameters are no-ops on frozen weights)

    # frozen-model conveniences the reference's callers touch
    def eval(self):
        return self

    def to(self, device):
        return self

    def parameters(self):
        return iter(())

    def state_dict(self):
        return self._state_dict

    @staticmethod
    def _layout(images):
        """visual_encoder.py:65-75: returns (tensor, nhwc, pre_scale)."""
        pre_scale = 1.0
        if isinstance(images, np.ndarray):
            # `torch.from_numpy(images).float() / 255.0` (:66): keep the raw values, fold the 1/255 into the kernel
            images = torch.from_numpy(np.ascontiguousarray(images))
            if images.dtype != torch.uint8:
                images = images.float()
            pre_scale = 1.0 / 255.0
        if images.dim() == 5:                    # [B, T, H, W, C]
            B, T, H, W, Cc = images.shape
            images = images.reshape(B * T, H, W, Cc)
            return images, True, pre_scale
        if images.dim() == 4 and images.shape[-1] == 3:
            return images, True, pre_scale
        if images.dim() != 4:
            raise ValueError(f"expected 4-D or 5-D images, got shape {tuple(images.shape)}")
        return images, False, pre_scale

    def forward_many(self, image_batches: Sequence, norm_mode: int = L.IMGNORM_AUTO) -> torch.Tensor:
        """Several reference `forward` calls (one per camera) in one engine launch sequence: each batch keeps its
        own normalisation decision.  Returns [n, B, hidden]."""
        laid = [self._layout(im) for im in image_batches]
        nhwc, pre = laid[0][1], laid[0][2]
        if any(l[1] != nhwc or l[2] != pre for l in laid):
            raise ValueError("all image batches of one call must share layout and dtype")
        tens = []
        for t, _, _ in laid:
            if t.dtype not in (torch.uint8, torch.float32):
                t = t.float()
            tens.append(t)
        if any(t.dtype != tens[0].dtype for t in tens):     # the engine reads every batch with ONE element type
            tens = [t.float() if t.dtype != torch.float32 else t for t in tens]
        out = self.engine.forward(tens, nhwc=nhwc, pre_scale=pre, norm_mode=norm_mode)
        if self._range is not None and not self._range.fell_back:
            bits = self._range.after(self.engine, self.engine.adt == L.F16)
            if bits:
                self._range.fall_back(bits)
                self.engine = DinoEngine(self._state_dict, heads=self._heads, precision="bf16", device=self.device)
                out = self.engine.forward(tens, nhwc=nhwc, pre_scale=pre, norm_mode=norm_mode)
        return out

    def forward(self, images):
        with torch.no_grad():
            return self.forward_many([images])[0]

    __call__ = forward
