"""Oracle: conditional 1-D U-Net forward (test infrastructure; see oracle/__init__.py).

Restates /root/reference/VLA/residual_controller/bridge/networks/conditional_unet_1D.py
(SinusoidalPosEmb :7-19, Conv1dBlock :40-55, ConditionalResidualBlock1D :58-105,
DiffusionConditionalUnet1D.forward :194-247) functionally over a state dict.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def mish(x: torch.Tensor) -> torch.Tensor:
    return x * torch.tanh(F.softplus(x))


def sinusoidal_pos_emb(t: torch.Tensor, dim: int) -> torch.Tensor:
    """conditional_unet_1D.py:12-19 — cat(sin, cos) of t * exp(-k*ln(1e4)/(half-1))."""
    half = dim // 2
    c = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -c)
    e = t.float()[:, None] * freqs[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def conv1d_block(sd: SD, p: str, x: torch.Tensor, n_groups: int = 8) -> torch.Tensor:
    """Conv1d(pad=k//2) -> GroupNorm -> Mish (conditional_unet_1D.py:48-55)."""
    w = sd[f"{p}.block.0.weight"]
    y = F.conv1d(x, w, sd[f"{p}.block.0.bias"], padding=w.shape[-1] // 2)
    y = F.group_norm(y, n_groups, sd[f"{p}.block.1.weight"], sd[f"{p}.block.1.bias"], eps=1e-5)
    return mish(y)


def res_block(sd: SD, p: str, x: torch.Tensor, g: torch.Tensor, n_groups: int = 8) -> torch.Tensor:
    """conditional_unet_1D.py:87-105 — FiLM (scale, bias) from Mish->Linear of the global feature."""
    out = conv1d_block(sd, f"{p}.blocks.0", x, n_groups)
    cout = out.shape[1]
    emb = F.linear(mish(g), sd[f"{p}.cond_encoder.1.weight"], sd[f"{p}.cond_encoder.1.bias"])
    scale, bias = emb[:, :cout, None], emb[:, cout:, None]
    out = scale * out + bias
    out = conv1d_block(sd, f"{p}.blocks.1", out, n_groups)
    if f"{p}.residual_conv.weight" in sd:
        res = F.conv1d(x, sd[f"{p}.residual_conv.weight"], sd[f"{p}.residual_conv.bias"])
    else:
        res = x
    return out + res


def step_embedding(sd: SD, prefix: str, t: torch.Tensor, dsed: int = 256) -> torch.Tensor:
    """diffusion_step_encoder: sinusoid -> Linear -> Mish -> Linear (conditional_unet_1D.py:133-138)."""
    e = sinusoidal_pos_emb(t, dsed)
    e = F.linear(e, sd[f"{prefix}diffusion_step_encoder.1.weight"], sd[f"{prefix}diffusion_step_encoder.1.bias"])
    e = mish(e)
    return F.linear(e, sd[f"{prefix}diffusion_step_encoder.3.weight"], sd[f"{prefix}diffusion_step_encoder.3.bias"])


def unet_forward(sd: SD, prefix: str, sample: torch.Tensor, timestep: torch.Tensor,
                 global_cond: torch.Tensor, n_down: int = 3, n_groups: int = 8) -> torch.Tensor:
    """sample (B,T,C), timestep (B,), global_cond (B,G) -> (B,T,C)  (conditional_unet_1D.py:194-247)."""
    x = sample.movedim(-1, -2).float()
    t = timestep.expand(x.shape[0])
    g = torch.cat([step_embedding(sd, prefix, t), global_cond.float()], dim=-1)
    h = []
    for i in range(n_down):
        x = res_block(sd, f"{prefix}down_modules.{i}.0", x, g, n_groups)
        x = res_block(sd, f"{prefix}down_modules.{i}.1", x, g, n_groups)
        h.append(x)
        if f"{prefix}down_modules.{i}.2.conv.weight" in sd:      # Downsample1d: Conv1d(k3,s2,p1)
            x = F.conv1d(x, sd[f"{prefix}down_modules.{i}.2.conv.weight"],
                         sd[f"{prefix}down_modules.{i}.2.conv.bias"], stride=2, padding=1)
    for i in range(2):
        x = res_block(sd, f"{prefix}mid_modules.{i}", x, g, n_groups)
    for i in range(n_down - 1):
        x = torch.cat((x, h.pop()), dim=1)
        x = res_block(sd, f"{prefix}up_modules.{i}.0", x, g, n_groups)
        x = res_block(sd, f"{prefix}up_modules.{i}.1", x, g, n_groups)
        if f"{prefix}up_modules.{i}.2.conv.weight" in sd:        # Upsample1d: ConvTranspose1d(k4,s2,p1)
            x = F.conv_transpose1d(x, sd[f"{prefix}up_modules.{i}.2.conv.weight"],
                                   sd[f"{prefix}up_modules.{i}.2.conv.bias"], stride=2, padding=1)
    x = conv1d_block(sd, f"{prefix}final_conv.0", x, n_groups)
    x = F.conv1d(x, sd[f"{prefix}final_conv.1.weight"], sd[f"{prefix}final_conv.1.bias"])
    return x.movedim(-1, -2)
