"""Minimal parameter containers for the API mirrors (no torch.nn compute anywhere).

`ParamModule` keeps an ordered {key: tensor} map with the reference's state-dict key layout and the
handful of nn.Module methods the reference's callers use (state_dict / load_state_dict / parameters /
to / train / eval).  A version counter lets engines know when to re-pack frozen weights.
"""
from __future__ import annotations

import contextlib
import math
import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import torch


def default_precision() -> str:
    p = os.environ.get("VLATOUCH_PRECISION", "bf16")
    if p not in ("fp32", "bf16"):
        raise ValueError(f"VLATOUCH_PRECISION must be fp32 or bf16, got {p!r}")
    return p


def _init_tensor(key: str, shape: Sequence[int], gen: torch.Generator) -> torch.Tensor:
    """torch-default-like init (uniform +-1/sqrt(fan_in) for weights and biases; norm gains 1, shifts 0)."""
    shape = tuple(shape)
    leaf = key.split(".")[-1]
    if len(shape) >= 2:
        fan_in = int(torch.tensor(shape[1:]).prod())
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    if leaf in ("weight", "lambda1") and ("block.1" in key or "norm" in key or "layer_scale" in key or key.startswith("1.")):
        return torch.ones(shape)
    if leaf == "bias" and ("block.1" in key or "norm" in key):
        return torch.zeros(shape)
    return (torch.rand(shape, generator=gen) * 2 - 1) * 0.05


class ParamModule:
    def __init__(self, shapes: Mapping[str, Sequence[int]], device="cpu", seed: int = 0, materialize: bool = True):
        """`materialize=False` records the shapes only (zero-size placeholders): for billion-parameter models whose weights
        arrive through load_state_dict(..., assign=True)."""
        gen = torch.Generator().manual_seed(seed)
        self._params: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._shapes = OrderedDict((k, tuple(int(x) for x in s)) for k, s in shapes.items())
        for k, s in shapes.items():
            self._params[k] = _init_tensor(k, s, gen).to(device) if materialize else torch.empty(0)
        self.training = False
        self.version = 0

    # --- nn.Module-like surface used by the reference's callers
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v) for k, v in self._params.items())

    def load_state_dict(self, sd: Mapping[str, torch.Tensor], strict: bool = True, assign: bool = False):
        """`assign=True` (torch's keyword) adopts the given tensors as they are (device and dtype preserved, no copy) —
        used to load multi-GB weights that already live in HBM in their execution dtype."""
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                               f"unexpected keys {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        for k in self._params:
            if k in sd:
                t = torch.as_tensor(sd[k])
                if tuple(t.shape) != self._shapes[k]:
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {self._shapes[k]}")
                self._params[k] = t.detach() if assign else t.detach().to(self._params[k].device, torch.float32).clone()
        self.version += 1
        return self

    def named_parameters(self) -> Iterable[Tuple[str, torch.Tensor]]:
        return iter(self._params.items())

    def parameters(self) -> Iterable[torch.Tensor]:
        return iter(self._params.values())

    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            for key in self._params:
                self._params[key] = self._params[key].to(device)
            self.version += 1
        return self

    def cuda(self, *a, **k):
        return self.to("cuda")

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def requires_grad_(self, *a, **k):
        return self

    def sub_state(self, prefix: str) -> Dict[str, torch.Tensor]:
        n = len(prefix)
        return {k[n:]: v for k, v in self._params.items() if k.startswith(prefix)}


class ExponentialMovingAverage:
    """State-compatible stand-in for torch_ema.ExponentialMovingAverage (bridge_model.py:10,433,267):
    a flat list of shadow parameters in `parameters()` order; `average_parameters()` swaps them in."""

    def __init__(self, parameters: Iterable[torch.Tensor], decay: float, use_num_updates: bool = True):
        self._owner: Optional[ParamModule] = None
        params = list(parameters)
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params: List[torch.Tensor] = [p.detach().clone() for p in params]
        self.collected_params = None
        self.version = 0

    def bind(self, owner: ParamModule):
        self._owner = owner
        return self

    def to(self, device=None, dtype=None):
        self.shadow_params = [p.to(device=device) for p in self.shadow_params]
        return self

    def update(self, parameters: Optional[Iterable[torch.Tensor]] = None):
        params = list(parameters) if parameters is not None else list(self._owner.parameters())
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        for s, p in zip(self.shadow_params, params):
            s.sub_((1.0 - decay) * (s - p.to(s.device)))
        self.version += 1

    @contextlib.contextmanager
    def average_parameters(self, parameters=None):
        owner = self._owner
        if owner is None:
            yield
            return
        keys = list(owner._params.keys())
        saved = [owner._params[k] for k in keys]
        for k, s in zip(keys, self.shadow_params):
            owner._params[k] = s
        owner.version += 1
        try:
            yield
        finally:
            for k, s in zip(keys, saved):
                owner._params[k] = s
            owner.version += 1

    def state_dict(self) -> dict:
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow_params,
                "collected_params": self.collected_params}

    def load_state_dict(self, sd: Mapping) -> None:
        self.decay = sd["decay"]
        self.num_updates = sd["num_updates"]
        sp = list(sd["shadow_params"])
        if len(sp) != len(self.shadow_params):
            raise ValueError(f"shadow_params must have the same length as the parameters ({len(sp)} vs {len(self.shadow_params)})")
        dev = self.shadow_params[0].device if self.shadow_params else "cpu"
        self.shadow_params = [torch.as_tensor(p).detach().to(dev, torch.float32).clone() for p in sp]
        self.collected_params = sd.get("collected_params")
        self.version += 1
