"""Oracle: RDT transformer forward + RDTRunner sampling (test infrastructure; see oracle/__init__.py).

Restates /root/reference/VLA/models/rdt/blocks.py (TimestepEmbedder :28-66, CrossAttention
:72-138, RDTBlock :144-183, FinalLayer :186-202), models/rdt/model.py:126-165 (RDT.forward) and
models/rdt_runner.py (adaptors :88-120, conditional_sample :122-165, predict_action :225-250).

Third-party arithmetic restated here because the packages are absent and un-pinned by the
reference: timm `Attention` / `Mlp` / `RmsNorm` (RDT-1B upstream pins timm==1.0.3)
and diffusers `DPMSolverMultistepScheduler` (oracle/dpm_solver.py).  `rms_mode`:
  "meansq": x * rsqrt(mean(x^2) + eps) * w   (textbook RMSNorm; the default)
  "var"   : x * rsqrt(var_unbiased(x) + eps) * w  (what timm<=1.0.8 `rms_norm` computed)
The same switch exists in the HIP path.  Runs in `dtype` (bf16 = the
reference's execution dtype, rounding after every op like torch CPU does; fp32 = exact math).
"""
from __future__ import annotations

import math
import re
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dpm_solver

SD = Dict[str, torch.Tensor]


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6, mode: str = "meansq") -> torch.Tensor:
    if mode == "meansq":
        v = x.pow(2).mean(dim=-1, keepdim=True)
    elif mode == "var":
        v = torch.var(x, dim=-1, keepdim=True)
    else:
        raise ValueError(mode)
    return x * torch.rsqrt(v + eps) * w


def timestep_embedding(t: torch.Tensor, dim: int, dtype, max_period: float = 10000.0) -> torch.Tensor:
    """blocks.py:41-61 — cat(cos, sin), fp32 then cast."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(dtype)


def embed_mlp(sd: SD, p: str, t: torch.Tensor, dtype) -> torch.Tensor:
    e = timestep_embedding(t, 256, dtype)
    e = F.silu(F.linear(e, sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"]))
    return F.linear(e, sd[f"{p}.mlp.2.weight"], sd[f"{p}.mlp.2.bias"])


def _sdpa(q, k, v, mask=None):
    """softmax(q k^T / sqrt(hd) [+mask]) v in fp32 accumulate, result cast back (SDPA math path)."""
    dt = q.dtype
    s = (q.float() @ k.float().transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    return (torch.softmax(s, dim=-1) @ v.float()).to(dt)


def self_attention(sd: SD, p: str, x: torch.Tensor, heads: int, rms_mode: str) -> torch.Tensor:
    """timm Attention(qkv_bias=True, qk_norm=True, norm_layer=RmsNorm) (blocks.py:150-153)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[f"{p}.qkv.weight"], sd[f"{p}.qkv.bias"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    q = rms_norm(q, sd[f"{p}.q_norm.weight"], 1e-6, rms_mode)
    k = rms_norm(k, sd[f"{p}.k_norm.weight"], 1e-6, rms_mode)
    o = _sdpa(q, k, v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])


def cross_attention(sd: SD, p: str, x: torch.Tensor, c: torch.Tensor, mask, heads: int, rms_mode: str) -> torch.Tensor:
    """blocks.py:102-138."""
    B, N, C = x.shape
    L = c.shape[1]
    hd = C // heads
    q = F.linear(x, sd[f"{p}.q.weight"], sd[f"{p}.q.bias"]).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    kv = F.linear(c, sd[f"{p}.kv.weight"], sd[f"{p}.kv.bias"]).reshape(B, L, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv.unbind(0)
    q = rms_norm(q, sd[f"{p}.q_norm.weight"], 1e-6, rms_mode)
    k = rms_norm(k, sd[f"{p}.k_norm.weight"], 1e-6, rms_mode)
    m = None if mask is None else mask.reshape(B, 1, 1, L).expand(-1, -1, N, -1)
    o = _sdpa(q, k, v, m).permute(0, 2, 1, 3).reshape(B, N, C)
    return F.linear(o, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])


def ffn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """timm Mlp with tanh-GELU (blocks.py:157-161)."""
    return F.linear(F.gelu(F.linear(x, sd[f"{p}.fc1.weight"], sd[f"{p}.fc1.bias"]), approximate="tanh"),
                    sd[f"{p}.fc2.weight"], sd[f"{p}.fc2.bias"])


def rdt_forward(sd: SD, x, freq, t, lang_c, img_c, lang_mask=None, img_mask=None, *, heads: int,
                horizon: int, rms_mode: str = "meansq", prefix: str = "model.") -> torch.Tensor:
    """models/rdt/model.py:126-165."""
    dtype = x.dtype
    P = prefix
    te = embed_mlp(sd, f"{P}t_embedder", t, dtype).unsqueeze(1)
    fe = embed_mlp(sd, f"{P}freq_embedder", freq, dtype).unsqueeze(1)
    if te.shape[0] == 1:
        te = te.expand(x.shape[0], -1, -1)
    x = torch.cat([te, fe, x], dim=1)
    x = x + sd[f"{P}x_pos_embed"]
    lang_c = lang_c + sd[f"{P}lang_cond_pos_embed"][:, :lang_c.shape[1]]
    img_c = img_c + sd[f"{P}img_cond_pos_embed"]
    conds, masks = [lang_c, img_c], [lang_mask, img_mask]
    i = 0
    while f"{P}blocks.{i}.norm1.weight" in sd:
        b = f"{P}blocks.{i}"
        c, m = conds[i % 2], masks[i % 2]
        x = self_attention(sd, f"{b}.attn", rms_norm(x, sd[f"{b}.norm1.weight"], 1e-6, rms_mode), heads, rms_mode) + x
        x = cross_attention(sd, f"{b}.cross_attn", rms_norm(x, sd[f"{b}.norm2.weight"], 1e-6, rms_mode), c, m, heads, rms_mode) + x
        x = ffn(sd, f"{b}.ffn", rms_norm(x, sd[f"{b}.norm3.weight"], 1e-6, rms_mode)) + x
        i += 1
    x = rms_norm(x, sd[f"{P}final_layer.norm_final.weight"], 1e-6, rms_mode)
    x = ffn(sd, f"{P}final_layer.ffn_final", x)
    return x[:, -horizon:]


def adaptor(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """`linear` or `mlpNx_gelu` (tanh) projector (rdt_runner.py:88-106)."""
    if f"{p}.weight" in sd:
        return F.linear(x, sd[f"{p}.weight"], sd[f"{p}.bias"])
    i = 0
    while f"{p}.{i}.weight" in sd:
        if i > 0:
            x = F.gelu(x, approximate="tanh")
        x = F.linear(x, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"])
        i += 2
    return x


def predict_action(sd: SD, lang_tokens, lang_attn_mask, img_tokens, state_tokens, action_mask, ctrl_freqs,
                   x_init: torch.Tensor, *, heads: int, horizon: int, num_inference_steps: int = 5,
                   num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2",
                   prediction_type: str = "sample", rms_mode: str = "meansq", record: bool = False):
    """rdt_runner.py:225-250 + :122-165.  `x_init` replaces the `torch.randn` draw (:137-139)."""
    dtype = state_tokens.dtype
    st = torch.cat([state_tokens, action_mask], dim=2)
    lang_c = adaptor(sd, "lang_adaptor", lang_tokens)
    img_c = adaptor(sd, "img_adaptor", img_tokens)
    state_traj = adaptor(sd, "state_adaptor", st)
    noisy = x_init.to(dtype)
    am = action_mask.expand(-1, horizon, -1)
    sched = dpm_solver.DPMSolverPP2M(num_train_timesteps, beta_schedule, prediction_type)
    sched.set_timesteps(num_inference_steps)
    traj = [noisy]
    for t in sched.timesteps:
        a = adaptor(sd, "state_adaptor", torch.cat([noisy, am], dim=2))
        sa = torch.cat([state_traj, a], dim=1)
        out = rdt_forward(sd, sa, ctrl_freqs, torch.tensor([int(t)]), lang_c, img_c, lang_mask=lang_attn_mask,
                          heads=heads, horizon=horizon, rms_mode=rms_mode)
        noisy = sched.step(out, noisy).to(dtype)          # rdt_runner.py:158-160
        traj.append(noisy)
    noisy = noisy * am
    return (noisy, traj) if record else noisy
