"""RDT transformer — mirror of the reference's VLA/models/rdt/model.py:22-165 (`RDT`).

Same constructor arguments, parameter names (checkpoint compatible: `t_embedder.mlp.0.weight`, `blocks.{i}.attn.qkv.weight`,
`blocks.{i}.cross_attn.kv.weight`, `final_layer.ffn_final.fc2.weight`, ...), sin-cos initialised
position tables and zero-initialised final projection, and the same `forward(x, freq, t, lang_c, img_c, lang_mask=None,
img_mask=None)` contract.  The forward pass is the HIP driver behind vt_rdt_forward.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional

import torch

from vlatouch import synth
from vlatouch.module import ParamModule
from models.rdt.blocks import get_1d_sincos_pos_embed_from_grid, get_multimodal_cond_pos_embed


class RDT(ParamModule):
    """Robotics Diffusion Transformer (model.py:22)."""

    def __init__(self, output_dim=128, horizon=32, hidden_size=1152, depth=28, num_heads=16, max_lang_cond_len=1024,
                 img_cond_len=4096, lang_pos_embed_config=None, img_pos_embed_config=None, dtype=torch.bfloat16,
                 rms_mode: str = "meansq", init_weights: bool = True):
        self.output_dim, self.horizon, self.hidden_size, self.depth, self.num_heads = output_dim, horizon, hidden_size, depth, num_heads
        self.max_lang_cond_len, self.img_cond_len = max_lang_cond_len, img_cond_len
        self.dtype = dtype
        self.rms_mode = rms_mode
        self.lang_pos_embed_config, self.img_pos_embed_config = lang_pos_embed_config, img_pos_embed_config
        shapes = synth.rdt_runner_shapes(hidden=hidden_size, depth=depth, heads=num_heads, horizon=horizon, action_dim=output_dim,
                                         lang_token_dim=16, img_token_dim=16, state_token_dim=output_dim,
                                         max_lang_cond_len=max_lang_cond_len, img_cond_len=img_cond_len)
        shapes = OrderedDict((k[len("model."):], v) for k, v in shapes.items() if k.startswith("model."))
        super().__init__(shapes, device="cpu", seed=11, materialize=init_weights)
        if init_weights:
            self.initialize_weights()
        self._engine = None
        self._engine_version = -1

    def initialize_weights(self):
        """model.py:67-124: xavier-uniform Linears with zero bias, unit RMSNorm gains, sin-cos position tables, N(0, 0.02)
        embedder MLPs, zero final projection."""
        gen = torch.Generator().manual_seed(11)
        P = self._params
        for k, v in P.items():
            if v.dim() == 2:
                bound = math.sqrt(6.0 / (v.shape[0] + v.shape[1]))
                P[k] = (torch.rand(v.shape, generator=gen) * 2 - 1) * bound
            elif k.endswith(".bias"):
                P[k] = torch.zeros_like(v)
            elif v.dim() == 1:
                P[k] = torch.ones_like(v)
        D = self.hidden_size
        x_pos = get_multimodal_cond_pos_embed(D, OrderedDict([('timestep', 1), ('ctrl_freq', 1), ('state', 1), ('action', self.horizon)]))
        P["x_pos_embed"] = torch.from_numpy(x_pos).float().unsqueeze(0)
        if self.lang_pos_embed_config is None:
            lp = get_1d_sincos_pos_embed_from_grid(D, torch.arange(self.max_lang_cond_len).numpy())
        else:
            lp = get_multimodal_cond_pos_embed(D, OrderedDict(self.lang_pos_embed_config), embed_modality=False)
        P["lang_cond_pos_embed"] = torch.from_numpy(lp).float().unsqueeze(0)
        if self.img_pos_embed_config is None:
            ip = get_1d_sincos_pos_embed_from_grid(D, torch.arange(self.img_cond_len).numpy())
        else:
            ip = get_multimodal_cond_pos_embed(D, OrderedDict(self.img_pos_embed_config), embed_modality=False)
        P["img_cond_pos_embed"] = torch.from_numpy(ip).float().unsqueeze(0)
        for e in ("t_embedder", "freq_embedder"):
            for i in (0, 2):
                P[f"{e}.mlp.{i}.weight"] = torch.randn(P[f"{e}.mlp.{i}.weight"].shape, generator=gen) * 0.02
        P["final_layer.ffn_final.fc2.weight"] = torch.zeros_like(P["final_layer.ffn_final.fc2.weight"])
        P["final_layer.ffn_final.fc2.bias"] = torch.zeros_like(P["final_layer.ffn_final.fc2.bias"])
        self.version += 1

    def _standalone_engine(self, device):
        """Engine for a bare RDT.forward (no adaptors in play: they get zero placeholders)."""
        from vlatouch.rdt_engine import RdtEngine
        if self._engine is None or self._engine_version != self.version:
            D = self.hidden_size
            sd = {"model." + k: v for k, v in self.state_dict().items()}
            for name, kin in (("lang_adaptor", 16), ("img_adaptor", 16), ("state_adaptor", 2 * self.output_dim)):
                sd[f"{name}.weight"] = torch.zeros(D, kin)
                sd[f"{name}.bias"] = torch.zeros(D)
            self._engine = RdtEngine(sd, hidden=D, depth=self.depth, heads=self.num_heads, horizon=self.horizon, action_dim=self.output_dim,
                                     lang_token_dim=16, img_token_dim=16, state_token_dim=self.output_dim,
                                     max_lang_cond_len=self.max_lang_cond_len, img_cond_len=self.img_cond_len, lang_adaptor="linear",
                                     img_adaptor="linear", state_adaptor="linear", dtype=self.dtype, rms_mode=self.rms_mode, device=device)
            self._engine_version = self.version
        return self._engine

    def forward(self, x, freq, t, lang_c, img_c, lang_mask=None, img_mask=None):
        """x (B, horizon+1, D); freq (B,); t (B,) or (1,); lang_c (B, L, D); img_c (B, img_cond_len, D);
        lang_mask (B, L) bool, True = valid -> (B, horizon, output_dim)  (model.py:126-165)."""
        if img_mask is not None:
            raise NotImplementedError("img_mask is never passed by the reference's callers and is not implemented")
        dev = x.device if x.device.type == "cuda" else torch.device("cuda")
        with torch.no_grad():
            return self._standalone_engine(dev).forward(x, freq, t, lang_c, img_c, lang_mask)

    __call__ = forward
