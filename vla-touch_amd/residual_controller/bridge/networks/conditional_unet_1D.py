"""Conditional 1-D U-Net — mirror of the reference's
VLA/residual_controller/bridge/networks/conditional_unet_1D.py:108-247 (`DiffusionConditionalUnet1D`).

Same constructor, same state-dict key layout (checkpoint compatible) and the same call
contract `net(sample[B,T,C], timestep, global_cond=[B,G]) -> [B,T,C]`; the forward is the HIP U-Net engine
(vt_unet_forward): channel-last implicit-GEMM convolutions on MFMA, GroupNorm+Mish+FiLM fused over split-K
slabs.  No torch.nn compute.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from vlatouch import synth
from vlatouch.engine import UNetEngine
from vlatouch.module import ParamModule, default_precision


class DiffusionConditionalUnet1D(ParamModule):
    def __init__(self, input_dim, global_cond_dim, diffusion_step_embed_dim=256, down_dims=[256, 512, 1024],
                 kernel_size=5, n_groups=8, device="cpu", precision: Optional[str] = None, seed: int = 0):
        self.input_dim, self.global_cond_dim = input_dim, global_cond_dim
        self.dsed, self.down_dims, self.kernel_size, self.n_groups = diffusion_step_embed_dim, list(down_dims), kernel_size, n_groups
        self.precision = precision or default_precision()
        super().__init__(synth.unet_shapes(input_dim, global_cond_dim, diffusion_step_embed_dim, down_dims, kernel_size),
                         device=device, seed=seed)
        self._engine: Optional[UNetEngine] = None
        self._engine_version = -1

    def engine_kwargs(self):
        return dict(input_dim=self.input_dim, global_cond_dim=self.global_cond_dim, dsed=self.dsed, down_dims=self.down_dims,
                    kernel_size=self.kernel_size, n_groups=self.n_groups, precision=self.precision)

    def _get_engine(self, device) -> UNetEngine:
        if self._engine is None or self._engine_version != self.version:
            self._engine = UNetEngine([self.state_dict()], device=device, **self.engine_kwargs())
            self._engine_version = self.version
        return self._engine

    def forward(self, sample: torch.Tensor, timestep, global_cond=None):
        """x: (B,T,input_dim); timestep: (B,) tensor, 0-d tensor or number; global_cond: (B,global_cond_dim)."""
        if global_cond is None:
            raise NotImplementedError("the interpolant controller always passes global_cond")
        dev = sample.device if sample.device.type == "cuda" else torch.device("cuda")
        if torch.is_tensor(timestep) and timestep.numel() == 1:
            timestep = float(timestep)
        with torch.no_grad():
            return self._get_engine(dev).forward(sample, timestep, global_cond)[0]

    __call__ = forward
