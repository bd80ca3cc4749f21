"""Oracle: DINOv2 CLS-feature encoder (test infrastructure; see oracle/__init__.py).

Restates /root/reference/VLA/residual_controller/visual_encoder.py:56-106 (layout fix-ups, the
data-dependent `/255` and ImageNet-normalise branches, `pooler_output`) and the HF
`transformers.Dinov2Model` forward it calls (transformers 5.15 in the build container:
models/dinov2/modeling_dinov2.py — embeddings :57-116, layer :356-382, model :461-476):
patch-embed conv 14/14, CLS, bicubic-interpolated position embeddings, 12x[LN -> MHA ->
LayerScale -> +, LN -> MLP(GELU erf) -> LayerScale -> +], final LN, CLS row.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def prepare_images(images) -> torch.Tensor:
    """visual_encoder.py:65-81,95-106.  NB both branches look at the WHOLE call's batch."""
    if isinstance(images, np.ndarray):
        images = torch.from_numpy(images).float() / 255.0
    if images.dim() == 5:
        B, T, H, W, C = images.shape
        images = images.reshape(B * T, H, W, C).permute(0, 3, 1, 2)
    elif images.dim() == 4 and images.shape[-1] == 3:
        images = images.permute(0, 3, 1, 2)
    if images.max() > 1.0:
        images = images / 255.0
    if images.mean() < 0.5:
        return images
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return (images - mean) / std


def interpolate_pos_embed(pos: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    """modeling_dinov2.py:57-95: bicubic, align_corners=False, fp32; identity when the grid matches."""
    n_pos = pos.shape[1] - 1
    if gh * gw == n_pos and gh == gw:
        return pos
    s = int(round(n_pos ** 0.5))
    D = pos.shape[-1]
    patch = pos[:, 1:].reshape(1, s, s, D).permute(0, 3, 1, 2).float()
    patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, gh * gw, D)
    return torch.cat((pos[:, :1], patch), dim=1)


def dinov2_forward(sd: SD, pixel_values: torch.Tensor, heads: int, patch: int = 14, eps: float = 1e-6,
                   return_tokens: bool = False) -> torch.Tensor:
    """pixel_values (B,3,H,W) fp32 -> pooler_output (B,D)."""
    x = pixel_values.float()
    B, _, H, W = x.shape
    w = sd["embeddings.patch_embeddings.projection.weight"]
    D = w.shape[0]
    tok = F.conv2d(x, w, sd["embeddings.patch_embeddings.projection.bias"], stride=patch).flatten(2).transpose(1, 2)
    tok = torch.cat((sd["embeddings.cls_token"].expand(B, -1, -1), tok), dim=1)
    tok = tok + interpolate_pos_embed(sd["embeddings.position_embeddings"], H // patch, W // patch)
    hd = D // heads
    i = 0
    while f"encoder.layer.{i}.norm1.weight" in sd:
        p = f"encoder.layer.{i}"
        h = F.layer_norm(tok, (D,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps)
        q = F.linear(h, sd[f"{p}.attention.attention.query.weight"], sd[f"{p}.attention.attention.query.bias"])
        k = F.linear(h, sd[f"{p}.attention.attention.key.weight"], sd[f"{p}.attention.attention.key.bias"])
        v = F.linear(h, sd[f"{p}.attention.attention.value.weight"], sd[f"{p}.attention.attention.value.bias"])
        N = tok.shape[1]
        q, k, v = (z.view(B, N, heads, hd).transpose(1, 2) for z in (q, k, v))
        a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, N, D)
        a = F.linear(a, sd[f"{p}.attention.output.dense.weight"], sd[f"{p}.attention.output.dense.bias"])
        tok = a * sd[f"{p}.layer_scale1.lambda1"] + tok
        h = F.layer_norm(tok, (D,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps)
        if f"{p}.mlp.weights_in.weight" in sd:      # dinov2-giant: HF Dinov2SwiGLUFFN — x1, x2 = weights_in(h).chunk(2); weights_out(silu(x1) * x2)
            x1, x2 = F.linear(h, sd[f"{p}.mlp.weights_in.weight"], sd[f"{p}.mlp.weights_in.bias"]).chunk(2, dim=-1)
            h = F.linear(F.silu(x1) * x2, sd[f"{p}.mlp.weights_out.weight"], sd[f"{p}.mlp.weights_out.bias"])
        else:
            h = F.gelu(F.linear(h, sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"]))
            h = F.linear(h, sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"])
        tok = h * sd[f"{p}.layer_scale2.lambda1"] + tok
        i += 1
    tok = F.layer_norm(tok, (D,), sd["layernorm.weight"], sd["layernorm.bias"], eps)
    return tok if return_tokens else tok[:, 0, :]


def encode(sd: SD, images, heads: int) -> torch.Tensor:
    """DINOv2Encoder.forward (visual_encoder.py:56-93)."""
    return dinov2_forward(sd, prepare_images(images), heads)
