"""Oracle: SigLIP vision tower -> image tokens (test infrastructure; see oracle/__init__.py).  SURVEY §8f-1.

Restates what /root/reference/VLA/models/multimodal_encoder/siglip_encoder.py:33-56 takes from HF
`transformers.SiglipVisionModel` (`last_hidden_state`, select_feature == 'patch'): patch-embed conv 14/14 ("valid": a
384-pixel side gives 27 patches), learned position embedding (no CLS token), N x [LN -> MHA (q/k/v/out with bias, 72-wide
heads) -> +, LN -> MLP (fc1 -> tanh-GELU -> fc2) -> +], post_layernorm.  The attention-pooling head (`pooler_output`) is not
on the RDT path.  Pinned by tests/golden/g11_siglip.npz (outputs of HF's own module on the synthetic weights).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def strip_prefix(sd: SD) -> SD:
    """transformers 4.x keys carry a `vision_model.` prefix, 5.x keys do not."""
    if any(k.startswith("vision_model.") for k in sd):
        return {k[len("vision_model."):]: v for k, v in sd.items() if k.startswith("vision_model.")}
    return sd


def siglip_forward(sd: SD, pixel_values: torch.Tensor, heads: int, patch: int = 14, eps: float = 1e-6) -> torch.Tensor:
    """pixel_values (B,3,H,W), already rescaled / normalised by the image processor -> last_hidden_state (B, N, D)."""
    sd = strip_prefix(sd)
    x = pixel_values.float()
    w = sd["embeddings.patch_embedding.weight"].float()
    D = w.shape[0]
    t = F.conv2d(x, w, sd["embeddings.patch_embedding.bias"].float(), stride=patch)         # [B, D, g, g]
    B = t.shape[0]
    t = t.flatten(2).transpose(1, 2) + sd["embeddings.position_embedding.weight"].float()[None]
    N = t.shape[1]
    hd = D // heads
    i = 0
    while f"encoder.layers.{i}.layer_norm1.weight" in sd:
        p = f"encoder.layers.{i}"
        g = lambda k: sd[f"{p}.{k}"].float()
        h = F.layer_norm(t, (D,), g("layer_norm1.weight"), g("layer_norm1.bias"), eps)
        q = F.linear(h, g("self_attn.q_proj.weight"), g("self_attn.q_proj.bias")).view(B, N, heads, hd).transpose(1, 2)
        k = F.linear(h, g("self_attn.k_proj.weight"), g("self_attn.k_proj.bias")).view(B, N, heads, hd).transpose(1, 2)
        v = F.linear(h, g("self_attn.v_proj.weight"), g("self_attn.v_proj.bias")).view(B, N, heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, N, D)
        t = t + F.linear(a, g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"))
        h = F.layer_norm(t, (D,), g("layer_norm2.weight"), g("layer_norm2.bias"), eps)
        h = F.gelu(F.linear(h, g("mlp.fc1.weight"), g("mlp.fc1.bias")), approximate="tanh")
        t = t + F.linear(h, g("mlp.fc2.weight"), g("mlp.fc2.bias"))
        i += 1
    return F.layer_norm(t, (D,), sd["post_layernorm.weight"].float(), sd["post_layernorm.bias"].float(), eps)
