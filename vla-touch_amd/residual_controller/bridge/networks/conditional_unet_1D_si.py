"""Three conditional U-Nets (b_net, v_net, s_net) — mirror of the reference's
VLA/residual_controller/bridge/networks/conditional_unet_1D_si.py:4-50.  State-dict keys are
`{b_net,v_net,s_net}.<unet key>` in that registration order, which is also the order of the EMA shadow list.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch

from vlatouch.module import ParamModule, default_precision
from .conditional_unet_1D import DiffusionConditionalUnet1D


class _SubNet:
    """View of one sub-net inside the joint parameter map; callable like the reference's sub-module."""

    def __init__(self, parent: "InterpolantsConditionalUnet1D", name: str):
        self._parent, self._name = parent, name
        self.training = False

    def state_dict(self):
        return self._parent.sub_state(self._name + ".")

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def __call__(self, sample, timestep, global_cond=None):
        return self._parent.forward_net(self._name, sample, timestep, global_cond)

    forward = __call__


class InterpolantsConditionalUnet1D(ParamModule):
    NETS = ("b_net", "v_net", "s_net")

    def __init__(self, input_dim, global_cond_dim, diffusion_step_embed_dim=256, down_dims=[256, 512, 512], kernel_size=5,
                 n_groups=8, device="cpu", precision: Optional[str] = None):
        self.precision = precision or default_precision()
        self.cfg = dict(input_dim=input_dim, global_cond_dim=global_cond_dim, dsed=diffusion_step_embed_dim,
                        down_dims=list(down_dims), kernel_size=kernel_size, n_groups=n_groups)
        protos = [DiffusionConditionalUnet1D(input_dim, global_cond_dim, diffusion_step_embed_dim, down_dims, kernel_size, n_groups,
                                             device="cpu", precision=self.precision, seed=i) for i in range(3)]
        shapes = OrderedDict()
        init = {}
        for name, net in zip(self.NETS, protos):
            for k, v in net.state_dict().items():
                shapes[f"{name}.{k}"] = tuple(v.shape)
                init[f"{name}.{k}"] = v
        super().__init__(shapes, device="cpu")
        self.load_state_dict(init)
        self.to(device)
        self.b_net, self.v_net, self.s_net = (_SubNet(self, n) for n in self.NETS)
        self._engines = {}

    def engine(self, nets, device):
        """Engine evaluating the named sub-nets together from the CURRENT parameters (e.g. ('v_net','s_net') while the
        EMA context has swapped the shadow weights in).  Cached per (nets, parameter version)."""
        from vlatouch.engine import UNetEngine
        key = tuple(nets)
        ent = self._engines.get(key)
        if ent is None or ent[0] != self.version:
            eng = UNetEngine([self.sub_state(n + ".") for n in nets], device=device, precision=self.precision, **self.cfg)
            self._engines = {k: v for k, v in self._engines.items() if v[0] == self.version}
            self._engines[key] = (self.version, eng)
            ent = self._engines[key]
        return ent[1]

    def forward_net(self, name, sample, timestep, global_cond):
        dev = sample.device if sample.device.type == "cuda" else torch.device("cuda")
        if torch.is_tensor(timestep) and timestep.numel() == 1:
            timestep = float(timestep)
        with torch.no_grad():
            return self.engine((name,), dev).forward(sample, timestep, global_cond)[0]
