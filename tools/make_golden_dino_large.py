#!/usr/bin/env python
"""Golden vectors for dinov2-large (hidden 1024, 24 layers, 16 heads), dinov2-giant (1536, 40 layers, 24 heads, SwiGLU FFN) and giant's first 4 blocks from the REFERENCE's DINOv2Encoder around HF Dinov2Model, imported read-only in
this container with the synthetic weight generator of tests/cases.py (only the OUTPUT is stored):

    python tools/make_golden_dino_large.py [large giant-l4 giant]     # writes tests/golden/g3_dino_{large,giant_l4,giant}.npz

Kept apart from tools/make_golden.py (same machinery, same reference class) so that the large model is only built when this one fixture is regenerated."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from tests import cases  # noqa: E402
import ref_import  # noqa: E402
from vlatouch import synth  # noqa: E402

torch.set_grad_enabled(False)
ref_import.setup()
ref_import.no_cuda()


def build_dino(name: str):
    """`name` as the reference passes it to Dinov2Model.from_pretrained: .../dinov2-large, .../dinov2-giant, or .../dinov2-giant-l4 (the first 4 of
    giant's 40 blocks: the reference class only looks for 'giant' in the name)."""
    from transformers import Dinov2Config, Dinov2Model
    size = "large" if "large" in name else ("giant-l4" if "giant-l4" in name else "giant")
    c = synth.DINOV2_CONFIGS[size]
    cfg = Dinov2Config(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                       image_size=518, patch_size=14, mlp_ratio=4, qkv_bias=True, layerscale_value=1.0,
                       hidden_act="gelu", layer_norm_eps=1e-6, use_swiglu_ffn=c.get("swiglu", False))
    m = Dinov2Model(cfg).eval()
    sd = m.state_dict()
    pre = "dinov2-" + size.split("-")[0] + "."
    m.load_state_dict({k: cases.T(synth.tensor(pre + k, tuple(v.shape), "")).to(v.dtype) for k, v in sd.items()})
    return m


ref_import.patch_dinov2(build_dino)
from visual_encoder import DINOv2Encoder  # noqa: E402  (reference)

which = sys.argv[1:] or ["large", "giant-l4", "giant"]
for size in which:
    enc = DINOv2Encoder(model_name=f"facebook/dinov2-{size}", device="cpu")
    assert enc.hidden_size == synth.DINOV2_CONFIGS[size]["hidden"]
    tag = size.replace("-", "_")
    out = {f"{tag}_224_bright": enc.forward(cases.frames(2, 224, "bright")).numpy(),
           f"{tag}_224_dark": enc.forward(cases.frames(2, 224, "dark")).numpy()}
    np.savez_compressed(os.path.join(cases.GOLDEN, f"g3_dino_{tag}.npz"), **out)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})
    del enc
