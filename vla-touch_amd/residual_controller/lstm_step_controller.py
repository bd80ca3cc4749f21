"""Tactile LSTM residual head — mirror of the reference's
VLA/residual_controller/lstm_step_controller.py:13-391 (`TactileLSTMController`, `load_lstm_controller`).

Same constructor, modules (`obs_encoder`, `force_encoder`, `lstm`, `output_head` with torch state-dict keys,
checkpoint `tactile_controller.pt`), stateful single-step `predict(obs_cond, vla_action,
force, initialize=False)` with carried `(hidden_state, cell_state)`, `predict_sequence`, `forward(batch_dict)`,
`encode_observation`, `encode_force`, `reset_state`, `save` / `load`.  One control tick is one call into the HIP
engine (vt_lstm_step): force MLP -> 2-layer LSTM cell -> head -> residual add; (de)normalisation by
vt_action_normalize.  (`load_lstm_controller` in the reference reads undefined globals; here it takes them as
keyword arguments with the reference script's values as defaults.)
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from vlatouch import _lib as L
from vlatouch import synth
from vlatouch.engine import LstmEngine, MlpEngine, concat_obs
from vlatouch.module import ParamModule, default_precision
from residual_controller.visual_encoder import DINOv2Encoder
from residual_controller.controller_dataset import denormalize_actions, normalize_actions
from residual_controller.bridge_controller import _Sequential


class _LSTMParams(ParamModule):
    def __init__(self, shapes, num_layers):
        super().__init__(shapes, device="cpu", seed=3)
        self.num_layers = num_layers


class TactileLSTMController:
    def __init__(self, state_dim=10, hidden_dim=256, num_layers=2, dropout=0.1, image_model_path="facebook/dinov2-small",
                 device="cuda", force_dim=3, use_force=True, precision: Optional[str] = None, image_state_dict=None):
        if hidden_dim not in (128, 256, 384) or state_dim > 16:
            # the inference head is ONE persistent kernel per call (csrc/vt_lstm.hip) that deals the 4 x hidden gate rows over its 8 waves in whole
            # 16-unit tiles: hidden_dim 128 / 256 (the reference's default, lstm_step_controller.py:16) / 384
            raise ValueError(f"TactileLSTMController on MI355X: hidden_dim must be 128, 256 or 384 and state_dim <= 16 (got {hidden_dim}, {state_dim})")
        self.state_dim = state_dim
        self.hidden_dim = hidden_dim
        self.device = device
        self.force_dim = force_dim
        self.use_force = use_force
        self.precision = precision or default_precision()
        self.image_encoder = DINOv2Encoder(model_name=image_model_path, device=device, precision=self.precision,
                                           state_dict=image_state_dict)
        self.latent_obs_dim = self.image_encoder.hidden_size
        shp = synth.lstm_controller_shapes(self.latent_obs_dim, state_dim, hidden_dim, num_layers, force_dim)
        self.obs_dim = self.latent_obs_dim * 2 + self.state_dim
        self.lstm_input_dim = hidden_dim // 2 + state_dim
        self.force_encoder = _Sequential(shp["force_encoder"], "fp32", device, seed=4)
        self.obs_encoder = _Sequential(shp["obs_encoder"], "fp32", device, seed=5)
        self.lstm = _LSTMParams(shp["lstm"], num_layers)
        self.output_head = _Sequential(shp["output_head"], "fp32", device, seed=6)
        self.use_residual = True
        self.hidden_state = None
        self.cell_state = None
        self.stats = None
        self.trainable_modules = [self.obs_encoder, self.force_encoder, self.lstm, self.output_head]
        self._engine: Optional[LstmEngine] = None
        self._engine_key = None

    def to(self, device):
        self.device = device
        for m in self.trainable_modules:
            m.to(device)
        return self

    def _lstm_engine(self) -> LstmEngine:
        key = tuple(m.version for m in self.trainable_modules)
        if self._engine is None or self._engine_key != key:
            mods = {"force_encoder": self.force_encoder.state_dict(), "lstm": self.lstm.state_dict(),
                    "output_head": self.output_head.state_dict()}
            self._engine = LstmEngine(mods, state_dim=self.state_dim, hidden=self.hidden_dim, layers=self.lstm.num_layers,
                                      force_dim=self.force_dim, precision="fp32" if self.precision == "fp32" else "x3", device=self.device)
            self._engine_key = key
        return self._engine

    def encode_images(self, images_cam1, images_cam2):
        feats = self.image_encoder.forward_many([images_cam1, images_cam2])
        return feats[0], feats[1]

    def encode_observation(self, state, images_cam1, images_cam2):
        """[cam1 | cam2 | state] -> obs MLP -> obs_cond [B, hidden] (lstm_step_controller.py:126-146; no force here)."""
        f1, f2 = self.encode_images(images_cam1, images_cam2)
        eng = self.obs_encoder.engine(torch.device(self.device))
        return eng(concat_obs(f1, f2, torch.as_tensor(state), None, eng.in_pad, eng.adt, torch.device(self.device)))

    def encode_force(self, force):
        """[B, T, force_dim] or [B, force_dim] -> [.., hidden//2] (lstm_step_controller.py:148-168)."""
        force = torch.as_tensor(force)
        if force.dim() == 3:
            B, T, _ = force.shape
            return self.force_encoder(force.reshape(B * T, -1)).reshape(B, T, -1)
        return self.force_encoder(force)

    def reset_state(self, batch_size=1):
        self.hidden_state = torch.zeros(self.lstm.num_layers, batch_size, self.hidden_dim, device=self.device)
        self.cell_state = torch.zeros(self.lstm.num_layers, batch_size, self.hidden_dim, device=self.device)

    def _step_n(self, obs_cond, vla_n, force):
        return self._lstm_engine().step(obs_cond, vla_n, force, self.hidden_state, self.cell_state)

    def forward(self, batch_dict):
        """Whole normalised sequence from a zero state -> vla_act + delta (lstm_step_controller.py:170-213)."""
        vla, obs_cond, forces = batch_dict['vla_act'], batch_dict['obs_cond'], batch_dict['forces']
        B, T, _ = vla.shape
        saved = (self.hidden_state, self.cell_state)
        self.reset_state(B)
        with torch.no_grad():
            o = self._lstm_engine().sequence(obs_cond, torch.as_tensor(vla), torch.as_tensor(forces), self.hidden_state, self.cell_state)
            if not self.use_residual:
                o = o - torch.as_tensor(vla).to(o.device)
        self.hidden_state, self.cell_state = saved
        return o

    def predict(self, obs_cond, vla_action, force, initialize=False):
        """One tick: normalised vla_action [B, state_dim] + force [B, force_dim] -> refined action in expert scale
        (lstm_step_controller.py:232-286)."""
        self.eval()
        with torch.no_grad():
            batch_size = vla_action.shape[0]
            if initialize or self.hidden_state is None:
                self.reset_state(batch_size)
            out_n = self._step_n(obs_cond, vla_action, force)
            if not self.use_residual:
                out_n = out_n - torch.as_tensor(vla_action).to(out_n.device)
            return denormalize_actions(out_n, self.stats, 'expert')

    def predict_sequence(self, obs_cond, vla_actions, force_seq):
        """lstm_step_controller.py:288-319."""
        B, T, _ = vla_actions.shape
        self.reset_state(batch_size=B)
        vla_actions_n = normalize_actions(torch.as_tensor(vla_actions).to(self.device), self.stats, 'vla')
        # the T ticks run as ONE persistent kernel with (h, c) carried on chip (vt_lstm_sequence) — same arithmetic as T calls of
        # `predict` (the reference's loop, :300-317)
        self.eval()
        with torch.no_grad():
            out_n = self._lstm_engine().sequence(obs_cond, vla_actions_n, torch.as_tensor(force_seq), self.hidden_state, self.cell_state)
            if not self.use_residual:
                out_n = out_n - vla_actions_n.to(out_n.device)
            return denormalize_actions(out_n, self.stats, 'expert')

    def get_loss(self, batch_dict):
        """F.mse_loss(forward(batch), expert_act) (lstm_step_controller.py:321-337) as a 0-d tensor.  Evaluation arithmetic (no
        dropout) through the sequence kernel; the differentiated loss with AdamW lives in `vlatouch.train.LstmTrainer` /
        `residual_controller.lstm_train.LSTMControllerTrainer`, which owns the optimiser state."""
        pred = self.forward(batch_dict)
        target = torch.as_tensor(batch_dict['expert_act']).to(pred.device, torch.float32)
        return ((pred - target) ** 2).mean()

    def train(self, mode=True):
        for m in self.trainable_modules:
            m.train(mode)
        return self

    def eval(self):
        for m in self.trainable_modules:
            m.eval()
        return self

    def save(self, path):
        cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}
        state_dict = {'stats': self.stats, 'model_args': getattr(self, 'model_args', None),
                      'modules': {'obs_encoder': cpu(self.obs_encoder.state_dict()), 'force_encoder': cpu(self.force_encoder.state_dict()),
                                  'lstm': cpu(self.lstm.state_dict()), 'output_head': cpu(self.output_head.state_dict())}}
        torch.save(state_dict, f"{path}/tactile_controller.pt")

    def load(self, path):
        checkpoint = torch.load(f"{path}/tactile_controller.pt", map_location="cpu", weights_only=False)
        modules = checkpoint['modules']
        self.obs_encoder.load_state_dict(modules['obs_encoder'])
        self.force_encoder.load_state_dict(modules['force_encoder'])
        self.lstm.load_state_dict(modules['lstm'])
        self.output_head.load_state_dict(modules['output_head'])
        self.stats = {key: torch.as_tensor(np.asarray(value), dtype=torch.float32).to(self.device)
                      for key, value in checkpoint['stats'].items()}
        self.model_args = checkpoint.get('model_args', None)


def load_lstm_controller(state_dim=10, force_dim=3, device="cuda", **kw):
    return TactileLSTMController(state_dim=state_dim, hidden_dim=256, num_layers=2, dropout=0.1, device=device,
                                 force_dim=force_dim, **kw)
