"""Oracle: DiffusionController.predict and TactileLSTMController (test infrastructure).

Restates /root/reference/VLA/residual_controller/bridge_controller.py:86-182 (encode_images,
encode_observation, predict) and lstm_step_controller.py:126-319 (encode_observation, forward,
predict, predict_sequence) over plain state dicts.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dinov2, interpolant, normalize, unet1d

SD = Dict[str, torch.Tensor]


def mlp3_gelu(sd: SD, x: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """Linear-GELU(erf)-Linear-GELU-Linear (bridge_controller.py:42-48)."""
    h = F.gelu(F.linear(x, sd[f"{prefix}0.weight"], sd[f"{prefix}0.bias"]))
    h = F.gelu(F.linear(h, sd[f"{prefix}2.weight"], sd[f"{prefix}2.bias"]))
    return F.linear(h, sd[f"{prefix}4.weight"], sd[f"{prefix}4.bias"])


def encode_observation(dino_sd: SD, heads: int, enc_sd: SD, state, cam1, cam2, forces=None) -> torch.Tensor:
    """bridge_controller.py:112-134: two SEPARATE DINO calls, cat(cam1, cam2, state[, forces]) -> MLP."""
    f1 = dinov2.encode(dino_sd, cam1, heads)
    f2 = dinov2.encode(dino_sd, cam2, heads)
    s = state.float()
    if forces is not None:
        s = torch.cat((s, forces.float()), dim=-1)
    return mlp3_gelu(enc_sd, torch.cat((f1, f2, s), dim=-1))


def predict(dino_sd: SD, heads: int, enc_sd: SD, net_sd: SD, stats, state, vla_actions, cam1, cam2,
            forces, noise, diffuse_step: int = 10, beta_max: float = 0.03, record: bool = False):
    """bridge_controller.py:149-182 with the sampler's weights = `net_sd` (the EMA shadow values)."""
    cond = encode_observation(dino_sd, heads, enc_sd, state, cam1, cam2, forces)
    x0 = normalize.normalize_actions(vla_actions.float(), stats, "vla")
    v = lambda x, t, c: unet1d.unet_forward(net_sd, "v_net.", x, t, c)
    s = lambda x, t, c: unet1d.unet_forward(net_sd, "s_net.", x, t, c)
    xT, traj = interpolant.sde_vs(v, s, x0, cond, noise, diffuse_step, beta_max)
    out = normalize.denormalize_actions(xT, stats, "expert")
    return (out, traj, cond) if record else out


# ---------------------------------------------------------------- LSTM residual head

def lstm_cell(sd: SD, layer: int, x: torch.Tensor, h: torch.Tensor, c: torch.Tensor):
    """torch.nn.LSTM cell, gate order i,f,g,o."""
    z = F.linear(x, sd[f"weight_ih_l{layer}"], sd[f"bias_ih_l{layer}"]) + \
        F.linear(h, sd[f"weight_hh_l{layer}"], sd[f"bias_hh_l{layer}"])
    i, f, g, o = z.chunk(4, dim=-1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def force_encoder(sd: SD, f: torch.Tensor) -> torch.Tensor:
    """lstm_step_controller.py:44-48."""
    return F.linear(F.gelu(F.linear(f, sd["0.weight"], sd["0.bias"])), sd["2.weight"], sd["2.bias"])


def output_head(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """Linear(512->256) -> LayerNorm -> GELU -> (Dropout, eval) -> Linear(256->10) (:75-81)."""
    h = F.linear(x, sd["0.weight"], sd["0.bias"])
    h = F.gelu(F.layer_norm(h, (h.shape[-1],), sd["1.weight"], sd["1.bias"], 1e-5))
    return F.linear(h, sd["4.weight"], sd["4.bias"])


def lstm_step(mods: Dict[str, SD], obs_cond, vla_n, force, h, c, layers: int = 2):
    """One `predict` tick before denormalisation (lstm_step_controller.py:232-286): returns
    (vla_n + delta, h', c') with h, c of shape (layers, B, hidden)."""
    x = torch.cat((force_encoder(mods["force_encoder"], force.float()), vla_n.float()), dim=-1)
    h2, c2 = [], []
    for l in range(layers):
        hl, cl = lstm_cell(mods["lstm"], l, x, h[l], c[l])
        h2.append(hl)
        c2.append(cl)
        x = hl                                   # inter-layer dropout inactive in eval
    delta = output_head(mods["output_head"], torch.cat((x, obs_cond.float()), dim=-1))
    return vla_n + delta, torch.stack(h2), torch.stack(c2)


def lstm_forward(mods: Dict[str, SD], obs_cond, vla_n_seq, force_seq, layers: int = 2, hidden: int = 256):
    """`forward` (:170-213): whole normalised sequence from zero state, residual add, no denorm."""
    B, T, _ = vla_n_seq.shape
    h = torch.zeros(layers, B, hidden)
    c = torch.zeros(layers, B, hidden)
    outs = []
    for t in range(T):
        o, h, c = lstm_step(mods, obs_cond, vla_n_seq[:, t], force_seq[:, t], h, c, layers)
        outs.append(o)
    return torch.stack(outs, dim=1)


def lstm_predict_sequence(mods: Dict[str, SD], stats, obs_cond, vla_actions, force_seq, layers: int = 2, hidden: int = 256):
    """`predict_sequence` (:288-319): normalise('vla') -> step loop -> denormalise('expert') per tick."""
    vn = normalize.normalize_actions(vla_actions.float(), stats, "vla")
    out_n = lstm_forward(mods, obs_cond, vn, force_seq, layers, hidden)
    return normalize.denormalize_actions(out_n, stats, "expert")


def lstm_encode_observation(dino_sd: SD, heads: int, obs_sd: SD, state, cam1, cam2) -> torch.Tensor:
    """lstm_step_controller.py:126-146 (no force in the observation MLP)."""
    f1 = dinov2.encode(dino_sd, cam1, heads)
    f2 = dinov2.encode(dino_sd, cam2, heads)
    return mlp3_gelu(obs_sd, torch.cat((f1, f2, state.float()), dim=-1))
