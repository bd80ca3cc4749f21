"""Action (de)normalisation — mirror of the reference's
VLA/residual_controller/controller_dataset.py:303-346 (`normalize_actions`) and :349-384
(`denormalize_actions`): padded (x1.4) min-max range mapped to [-1, 1]; the <1e-6 range guard exists in
`normalize_actions` only; unknown `action_type` raises ValueError.  The arithmetic runs in the HIP kernel
behind `vt_action_normalize` (include/vlatouch.h); CPU tensors are staged through the GPU and returned on CPU.
The episode dataset / data module of the reference file are training-side and out of scope.
"""
from __future__ import annotations

import torch

from vlatouch import _lib as L
from vlatouch.engine import action_normalize


def _select(stats, action_type):
    if action_type == "expert":
        return stats["action_mins"], stats["action_maxs"]
    if action_type == "vla":
        return stats["vla_mins"], stats["vla_maxs"]
    raise ValueError(f"Unknown action_type: {action_type}")


def _run(x, stats, action_type, padding_factor, denorm):
    mins, maxs = _select(stats, action_type)
    x = torch.as_tensor(x)
    src_device = x.device
    dev = src_device if src_device.type == "cuda" else L.require_gpu("cuda")
    out = action_normalize(x.to(dev), torch.as_tensor(mins), torch.as_tensor(maxs), denorm=denorm, padding_factor=padding_factor)
    return out.to(src_device)


def normalize_actions(actions, stats, action_type="expert", padding_factor=1.4):
    return _run(actions, stats, action_type, padding_factor, False)


def denormalize_actions(normalized_actions, stats, action_type="expert", padding_factor=1.4):
    return _run(normalized_actions, stats, action_type, padding_factor, True)
