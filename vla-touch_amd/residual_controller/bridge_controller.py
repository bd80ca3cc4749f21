"""Interpolant controller facade — mirror of the reference's
VLA/residual_controller/bridge_controller.py:10-273 (`DiffusionController`, `load_bridge_controller`).

Same constructor arguments, attributes (`stats`, `model_args`, `diffusion_steps`, `state_encoder`,
`force_decoder`, `image_encoder`, `diffusion_model`, `latent_obs_dim`, `obs_dim`), methods (`to`,
`encode_images`, `encode_observation`, `predict`, `train` / `eval`, `save`, `load`) and checkpoint files
(`controller.pt` + `bridge_model.pt`).  `predict` is:

    two DINOv2 CLS encodings (one engine call, per-camera normalisation decisions)   vt_dino_forward
    -> cat(cam1, cam2, state, forces) -> 3-layer GELU MLP                              vt_concat_obs, vt_mlp
    -> normalise(vla) -> n-step velocity-score SDE over v_net/s_net -> denormalise     vt_action_normalize, vt_si_sample

all on the current HIP stream with no host synchronisation (the reference's two `.max()/.mean()` host
syncs per camera and its timing prints are gone).  Extra keyword: `noise=` injects the sampler's N(0,1) draws.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from vlatouch import _lib as L
from vlatouch import synth
from vlatouch.engine import MlpEngine, concat_obs
from vlatouch.module import ParamModule, default_precision
from residual_controller.visual_encoder import DINOv2Encoder
from residual_controller.bridge.bridge_model import StochasticInterpolants
from residual_controller.controller_dataset import denormalize_actions, normalize_actions


class _Sequential(ParamModule):
    """Parameter holder with nn.Sequential(Linear, GELU, Linear, GELU, Linear) key layout ('0.weight', '2.weight', ...)."""

    def __init__(self, shapes, precision, device, seed):
        super().__init__(shapes, device="cpu", seed=seed)
        self.precision = precision
        self._device = device
        self._engine: Optional[MlpEngine] = None
        self._engine_version = -1

    def engine(self, device) -> MlpEngine:
        if self._engine is None or self._engine_version != self.version:
            self._engine = MlpEngine(self.state_dict(), act=L.ACT_GELU_ERF, precision=self.precision, device=device)
            self._engine_version = self.version
        return self._engine

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        dev = x.device if x.device.type == "cuda" else torch.device("cuda")
        eng = self.engine(dev)
        return eng(eng.pad_input(x))


class DiffusionController:
    """Refines VLA action chunks with the stochastic-interpolant model (bridge_controller.py:10)."""

    def __init__(self, state_dim=10, hidden_dim=256, image_model_path="facebook/dinov2-small", diffusion_steps=10,
                 device="cuda", model_args=None, use_force=True, force_dim=3, precision: Optional[str] = None,
                 image_state_dict=None):
        self.state_dim = state_dim
        self.hidden_dim = hidden_dim
        self.device = device
        self.diffusion_steps = diffusion_steps
        self.precision = precision or default_precision()
        self.image_encoder = DINOv2Encoder(model_name=image_model_path, device=device, precision=self.precision,
                                           state_dict=image_state_dict)
        self.latent_obs_dim = self.image_encoder.hidden_size
        self.use_force = use_force
        self.force_dim = force_dim
        self.model_args = model_args
        self.stats = None
        # the tiny observation MLP runs with fp32 weights in both modes (0.33 M parameters, M = batch)
        mlp_prec = "fp32"
        if self.use_force:
            self.obs_dim = self.latent_obs_dim * 2 + self.state_dim + self.force_dim
            self.state_encoder = _Sequential(synth.state_encoder_shapes(self.obs_dim, hidden_dim), mlp_prec, device, seed=1)
            self.force_decoder = _Sequential(synth.force_decoder_shapes(hidden_dim, force_dim), mlp_prec, device, seed=2)
        else:
            print("No force perception!!!!")
            self.obs_dim = self.latent_obs_dim * 2 + self.state_dim
            self.state_encoder = _Sequential(synth.state_encoder_shapes(self.obs_dim, hidden_dim), mlp_prec, device, seed=1)
        self.diffusion_model = StochasticInterpolants(precision=self.precision)
        if self.model_args:
            self.diffusion_model.load_model(self.model_args, device)
        self.to(device)

    def to(self, device):
        self.device = device
        self.state_encoder.to(device)
        if self.use_force:
            self.force_decoder.to(device)
        return self

    def encode_images(self, images_cam1, images_cam2):
        """Both cameras through DINOv2 -> (cam1_features, cam2_features), each [B, hidden] (bridge_controller.py:86-110)."""
        if images_cam1 is None or images_cam2 is None:
            return None
        feats = self.image_encoder.forward_many([images_cam1, images_cam2])
        return feats[0], feats[1]

    def encode_observation(self, state, images_cam1=None, images_cam2=None, forces=None):
        """state [B, state_dim] (+ images, forces) -> obs_cond [B, hidden_dim] (bridge_controller.py:112-134)."""
        cam1_features, cam2_features = self.encode_images(images_cam1, images_cam2)
        eng = self.state_encoder.engine(torch.device(self.device))
        x = concat_obs(cam1_features, cam2_features, torch.as_tensor(state), torch.as_tensor(forces) if self.use_force else None,
                       eng.in_pad, eng.adt, torch.device(self.device))
        return eng(x)

    def predict(self, state, vla_actions, images_cam1=None, images_cam2=None, forces=None, noise=None, obs_cond=None):
        """Refined actions [B, horizon, state_dim] in the expert action scale (bridge_controller.py:149-182).
        `obs_cond=` (extension) passes an observation encoding computed earlier by `encode_observation`, e.g. on a second
        HIP stream while the VLA chunk is still being generated."""
        self.eval()
        with torch.no_grad():
            if obs_cond is None:
                obs_cond = self.encode_observation(state, images_cam1, images_cam2, forces)
            vla_actions_n = normalize_actions(torch.as_tensor(vla_actions).to(self.device), self.stats, 'vla')
            refined_actions_n = self.diffusion_model.sample(x_prior=vla_actions_n, cond=obs_cond,
                                                            diffuse_step=self.diffusion_steps, noise=noise, _own_prior=True)
            return denormalize_actions(refined_actions_n, self.stats, 'expert')

    def train(self):
        self.state_encoder.train()
        self.diffusion_model.train()
        if self.use_force:
            self.force_decoder.train()
        return self

    def eval(self):
        self.state_encoder.eval()
        self.diffusion_model.eval()
        if self.use_force:
            self.force_decoder.eval()
        return self

    def save(self, path):
        """controller.pt = {state_encoder, model_args, stats[, force_decoder]} + bridge_model.pt (bridge_controller.py:196-222)."""
        cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}
        state_dict = {'state_encoder': cpu(self.state_encoder.state_dict()), 'model_args': self.model_args, 'stats': self.stats}
        if self.use_force:
            state_dict['force_decoder'] = cpu(self.force_decoder.state_dict())
        torch.save(state_dict, f"{path}/controller.pt")
        self.diffusion_model.save_model(path)

    def load(self, path):
        """Read the reference's checkpoint files (bridge_controller.py:226-244)."""
        checkpoint = torch.load(f"{path}/controller.pt", map_location="cpu", weights_only=False)
        self.state_encoder.load_state_dict(checkpoint['state_encoder'])
        if self.use_force:
            self.force_decoder.load_state_dict(checkpoint['force_decoder'])
        self.state_encoder.to(self.device)
        self.model_args = checkpoint['model_args']
        self.stats = {key: torch.as_tensor(np.asarray(value), dtype=torch.float32).to(self.device)
                      for key, value in checkpoint['stats'].items()}
        self.diffusion_model.load_model({**self.model_args, 'ckpt_path': path, 'pretrain': True}, self.device)


def load_bridge_controller(**overrides):
    """The reference's default controller (bridge_controller.py:246-273); keyword overrides are forwarded."""
    model_args = {
        'interpolant_type': 'linear', 'gamma_type': '2^0.5*t(t-1)', 'epsilon_type': '1-t', 'prior_policy': 'vla',
        'beta_max': 0.03, 'sde_type': 'vs', 'action_dim': 10, 'obs_dim': 256, 'obs_horizon': 1, 'net_type': 'unet1D_si',
        'pretrain': False, 'context_frames': 2, 'horizon': 16,
    }
    kw = dict(state_dim=10, hidden_dim=256, image_model_path="facebook/dinov2-small", diffusion_steps=10,
              model_args=model_args, force_dim=3)
    kw.update(overrides)
    return DiffusionController(**kw)
