"""Mirror of /root/reference/VLA/models/multimodal_encoder/siglip_encoder.py (SiglipVisionTower, :6-86): the RDT image tower.
Same constructor, `forward(images)` (tensor batch or list of single images -> last_hidden_state tokens), `feature_select`,
and properties (`hidden_size`, `num_patches`, `num_patches_per_side`, `dtype`, `device`, `config`, `dummy_feature`); the
transformer runs in the HIP engine (vlatouch.engine.SiglipEngine).  There is no hub access here: weights come from
`state_dict=`, a local HF directory (config.json + model.safetensors / pytorch_model.bin) given as `vision_tower`, or — with
VLATOUCH_SYNTH_WEIGHTS=1 — the deterministic synthetic set.  The image processor (resize / rescale / normalise, host-side PIL
work in the reference's `step`) is not part of this class's arithmetic; `image_processor` exposes its constants."""
from __future__ import annotations

import json
import os
import types
from typing import Dict, Optional

import torch

from vlatouch import _lib as L
from vlatouch import synth
from vlatouch.engine import AutoRange, SiglipEngine
from vlatouch.module import default_precision

_SO400M = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, image_size=384, patch_size=14)


def _load_local(path: str):
    cfg = dict(_SO400M)
    cj = os.path.join(path, "config.json")
    if os.path.isfile(cj):
        c = json.load(open(cj))
        c = c.get("vision_config", c)
        cfg.update({k: c[k] for k in cfg if k in c})
    st = os.path.join(path, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        return cfg, load_file(st)
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.isfile(pt):
        return cfg, torch.load(pt, map_location="cpu")
    return cfg, None


class SiglipVisionTower:
    def __init__(self, vision_tower, args=None, delay_load=False, *, device="cuda", precision: Optional[str] = None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, config: Optional[dict] = None):
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self._device = torch.device(device)
        self.precision = precision or default_precision()
        self._state_dict, self._cfg = state_dict, dict(_SO400M, **(config or {}))
        # SiglipImageProcessor constants (rescale 1/255, mean = std = 0.5, square resize to image_size)
        self.image_processor = types.SimpleNamespace(image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5], rescale_factor=1 / 255.0,
                                                     size={"height": self._cfg["image_size"], "width": self._cfg["image_size"]})
        if not delay_load or getattr(args, "unfreeze_mm_vision_tower", False):
            self.load_model()

    def load_model(self, device_map=None):
        if self.is_loaded:
            print('{} is already loaded, `load_model` called again, skipping.'.format(self.vision_tower_name))
            return
        sd = self._state_dict
        if sd is None and os.path.isdir(str(self.vision_tower_name)):
            cfg, sd = _load_local(self.vision_tower_name)
            self._cfg.update(cfg)
        if sd is None:
            if os.environ.get("VLATOUCH_SYNTH_WEIGHTS") == "1":
                c = self._cfg
                shapes = synth.siglip_shapes(c["hidden_size"], c["num_hidden_layers"], c["intermediate_size"], c["image_size"], c["patch_size"])
                sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, prefix="siglip.").items()}
            else:
                raise FileNotFoundError(
                    f"no SigLIP weights for {self.vision_tower_name!r}: pass state_dict=, point vision_tower at a local HF directory, "
                    "or set VLATOUCH_SYNTH_WEIGHTS=1 for deterministic synthetic weights (no network access here)")
        prec = self.precision
        if prec == "bf16":      # as for DINOv2: IEEE fp16 storage (fp32 residual stream) is the low-precision mode of the encoders
            prec = os.environ.get("VLATOUCH_SIGLIP_PRECISION") or "fp16"
            if "VLATOUCH_SIGLIP_PRECISION" not in os.environ:      # range guard (round 6): the fp16 default falls back to bf16 storage when the engine flags a non-finite feature
                self._range = AutoRange(f"SiglipVisionTower({self.vision_tower_name})")
        self._sd_loaded = sd
        self.engine = SiglipEngine(sd, heads=self._cfg["num_attention_heads"], precision=prec, device=self._device,
                                   patch=self._cfg["patch_size"])
        self.vision_tower = self
        self.is_loaded = True

    def eval(self):
        return self

    def feature_select(self, image_forward_outs):
        if self.select_feature == "patch":
            return image_forward_outs
        if self.select_feature == "cls_patch":
            raise NotImplementedError("pooler_output (SigLIP attention-pooling head) is not on the RDT path and is not built")
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    def _encode(self, x):
        out = self.engine.forward(x)
        rg = getattr(self, "_range", None)
        if rg is not None and not rg.fell_back:
            bits = rg.after(self.engine, self.engine.adt == L.F16)
            if bits:
                rg.fall_back(bits)
                self.engine = SiglipEngine(self._sd_loaded, heads=self._cfg["num_attention_heads"], precision="bf16", device=self._device,
                                           patch=self._cfg["patch_size"])
                out = self.engine.forward(x)
        return out

    @torch.no_grad()
    def forward(self, images):
        if isinstance(images, list):
            return [self.feature_select(self._encode(im.unsqueeze(0))).to(im.dtype) for im in images]
        return self.feature_select(self._encode(images)).to(images.dtype)

    __call__ = forward

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return {"fp32": torch.float32, "bf16": torch.bfloat16}[self.precision]

    @property
    def device(self):
        return self._device

    @property
    def config(self):
        return types.SimpleNamespace(**self._cfg)

    @property
    def hidden_size(self):
        return self._cfg["hidden_size"]

    @property
    def num_patches_per_side(self):
        return self._cfg["image_size"] // self._cfg["patch_size"]

    @property
    def num_patches(self):
        return (self._cfg["image_size"] // self._cfg["patch_size"]) ** 2
