"""Oracle: padded min-max action (de)normalisation (test infrastructure).

Restates /root/reference/VLA/residual_controller/controller_dataset.py:303-346 (normalize)
and :349-384 (denormalize).  Note the asymmetry kept from the reference: the <eps guard on the
range exists only in `normalize`.
"""
from __future__ import annotations

import torch


def _bounds(stats, action_type: str, padding_factor: float):
    if action_type == "expert":
        mins, maxs = stats["action_mins"], stats["action_maxs"]
    elif action_type == "vla":
        mins, maxs = stats["vla_mins"], stats["vla_maxs"]
    else:
        raise ValueError(f"Unknown action_type: {action_type}")
    padded_range = (maxs - mins) * padding_factor
    center = (mins + maxs) / 2
    return center - padded_range / 2, center + padded_range / 2


def normalize_actions(actions, stats, action_type: str = "expert", padding_factor: float = 1.4):
    pmin, pmax = _bounds(stats, action_type, padding_factor)
    rng = pmax - pmin
    rng = torch.where(rng < 1e-6, torch.ones_like(rng), rng)
    return 2.0 * (actions - pmin) / rng - 1.0


def denormalize_actions(normalized, stats, action_type: str = "expert", padding_factor: float = 1.4):
    pmin, pmax = _bounds(stats, action_type, padding_factor)
    return (normalized + 1.0) / 2.0 * (pmax - pmin) + pmin
