"""Mirror of the reference's `DiffusionControllerTrainer` (VLA/residual_controller/bridge_train.py:27) on the device training step.

Same constructor, `_prepare_batch_for_diffusion`, `_train_epoch`, `_evaluate`, `train`, `_save_checkpoint` / `load_checkpoint`
(:33-80, :105-164, :166-250, :267-378, :380-443, :490-530).  What differs, and why:

* the optimiser, EMA and backward pass are `vlatouch.train.SITrainer` (HIP kernels through the C ABI), not torch.autograd: AdamW over
  `diffusion_model.net` + `state_encoder` (:49-56), cosine LR (T_max 100000, eta_min lr/10, :59-61), `ema.update()` each step (:334);
* `_prepare_batch_for_diffusion` returns `obs_in` (the observation MLP's *input*, [cls_cam1 | cls_cam2 | state | force]) next to
  `obs_cond`, because the MLP is trained through the interpolant loss and its backward needs the input;
* `t` and `z` (the reference's `torch.rand` / `torch.randn_like` draws inside `get_loss`, bridge_model.py:190,199) may be passed in for
  reproducible steps; they are drawn on the device otherwise;
* TensorBoard / file logging are replaced by a plain list of per-step records (`self.history`): no tensorboard in this image.
"""
from __future__ import annotations

import math
import os
import shutil
from typing import Dict, Optional

import torch

from vlatouch import _lib as L
from vlatouch.engine import concat_obs
from vlatouch.train import SITrainer

from .controller_dataset import normalize_actions


class DiffusionControllerTrainer:
    def __init__(self, controller, data_module, learning_rate=1e-4, weight_decay=1e-6, checkpoint_dir='checkpoint/bridge_controller/',
                 device='cuda'):
        self.controller = controller
        self.device = device
        self.checkpoint_dir = checkpoint_dir
        os.makedirs(checkpoint_dir, exist_ok=True)
        dm = controller.diffusion_model
        self.trainer = SITrainer(dm.net.state_dict(), controller.state_encoder.state_dict(), gamma_type=dm.gamma_type,
                                 interpolant_type=dm.interpolant_type, beta_max=dm.d, lr=learning_rate, weight_decay=weight_decay,
                                 ema_decay=dm.ema.decay, device=device)
        self.base_lr, self.eta_min, self.t_max = learning_rate, learning_rate / 10, 100000
        self.sched_step = 0
        self.use_force = controller.use_force
        self.controller.stats = data_module.stats
        self.stats = {k: torch.as_tensor(v, dtype=torch.float32).to(device) for k, v in data_module.stats.items()}
        self.history = []
        self._dirty = False

    # ------------------------------------------------------------------------------------------------ batch preparation
    def _prepare_batch_for_diffusion(self, batch) -> Dict[str, torch.Tensor]:
        states, forces = batch['states'], batch['forces']
        cf = (self.controller.model_args or {}).get('context_frames', 2)
        current_state, current_forces, future_forces = states[:, cf - 1], forces[:, cf - 1], forces[:, cf:]
        vla_n = normalize_actions(batch['vla_actions'], self.stats, 'vla')
        expert_n = normalize_actions(batch['expert_actions'], self.stats, 'expert')
        cam1, cam2 = batch.get('images_cam1'), batch.get('images_cam2')
        if cam1 is None or cam2 is None:
            raise ValueError("images_cam1 / images_cam2 are required: the observation encoding concatenates both camera features")
        f1, f2 = self.controller.encode_images(cam1[:, -1], cam2[:, -1])
        dev = torch.device(self.device)
        obs_in = concat_obs(f1, f2, torch.as_tensor(current_state), torch.as_tensor(current_forces) if self.use_force else None,
                            self.trainer.mlp.kin, L.F32, dev)
        return {'obs_in': obs_in, 'expert_act': expert_n, 'vla_act': vla_n, 'forces': future_forces, 'current_force': current_forces}

    # ------------------------------------------------------------------------------------------------ steps
    def _lr(self) -> float:
        return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.sched_step / self.t_max)) / 2

    def _draws(self, batch_dict, t, z):
        x0 = batch_dict['vla_act']
        dev = torch.device(self.device)
        if t is None:
            t = torch.rand(x0.shape[0], device=dev)
        if z is None:
            z = torch.randn(tuple(x0.shape), device=dev)
        return t, z

    def train_step(self, batch, t: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None):
        """One iteration of the reference's loop body (:309-337): prepare, get_loss, backward, AdamW, EMA, scheduler."""
        bd = self._prepare_batch_for_diffusion(batch)
        t, z = self._draws(bd, t, z)
        self.trainer.lr = self._lr()
        loss, info = self.trainer.train_step(bd['obs_in'], bd['vla_act'], bd['expert_act'], t, z)
        self.sched_step += 1
        self._dirty = True
        return loss, info

    def eval_step(self, batch, t=None, z=None):
        bd = self._prepare_batch_for_diffusion(batch)
        t, z = self._draws(bd, t, z)
        return self.trainer.get_loss(bd['obs_in'], bd['vla_act'], bd['expert_act'], t, z, backward=False)

    def sync_controller(self) -> None:
        """Write the trained tensors back into the controller objects (net, EMA shadow, state_encoder), so that
        `controller.predict` / `controller.save` see them."""
        if not self._dirty:
            return
        dm = self.controller.diffusion_model
        dm.net.load_state_dict(self.trainer.net_state_dict())
        ema_sd = self.trainer.ema_state_dict()
        dm.ema.load_state_dict({"decay": dm.ema.decay, "num_updates": self.trainer.ema_updates,
                                "shadow_params": [ema_sd[k] for k in dm.net.state_dict().keys()], "collected_params": None})
        self.controller.state_encoder.load_state_dict(self.trainer.mlp.state_dict())
        dm._sampler = None
        self._dirty = False

    # ------------------------------------------------------------------------------------------------ epochs
    def _train_epoch(self, dataloader, global_step, log_interval, diffusion_steps_schedule, epoch):
        self.controller.train()
        total, n = 0.0, 0
        for batch_idx, batch in enumerate(dataloader):
            step = global_step + batch_idx
            if diffusion_steps_schedule is not None:
                self.controller.diffusion_steps = diffusion_steps_schedule(step)
            loss, info = self.train_step(batch)
            total, n = total + loss, n + 1
            if step % log_interval == 0:
                self.history.append({"step": step, "epoch": epoch, "loss": loss, **info, "lr": self.trainer.lr})
        return total / max(n, 1)

    def _evaluate(self, dataloader, epoch):
        self.controller.eval()
        total, n = 0.0, 0
        for batch in dataloader:
            loss, _ = self.eval_step(batch)
            total, n = total + loss, n + 1
        avg = total / max(n, 1)
        self.history.append({"epoch": epoch, "val_loss": avg})
        return avg

    def train(self, data_module, num_epochs=100, save_interval=25, eval_interval=5, log_interval=100, diffusion_steps_schedule=None):
        train_dl, val_dl = data_module.train_dataloader(), data_module.val_dataloader()
        global_step, best = 0, float('inf')
        for epoch in range(num_epochs):
            self._train_epoch(train_dl, global_step, log_interval, diffusion_steps_schedule, epoch + 1)
            global_step += len(train_dl)
            if (epoch + 1) % eval_interval == 0 or (epoch + 1) == 1:
                val = self._evaluate(val_dl, epoch)
                if val < best:
                    best = val
                    self._remove_checkpoint("best_model")
                    self._save_checkpoint("best_model")
            if (epoch + 1) % save_interval == 0 or (epoch + 1) == 1:
                self._remove_checkpoint(f"epoch_{epoch + 1 - save_interval}")
                self._save_checkpoint(f"epoch_{epoch + 1}")
        return best

    # ------------------------------------------------------------------------------------------------ checkpoints
    def _remove_checkpoint(self, checkpoint_name):
        path = os.path.join(self.checkpoint_dir, checkpoint_name)
        if os.path.isdir(path):
            shutil.rmtree(path)

    def _save_checkpoint(self, name):
        save_path = os.path.join(self.checkpoint_dir, name)
        os.makedirs(save_path, exist_ok=True)
        self.sync_controller()
        self.controller.save(save_path)

    def load_checkpoint(self, path):
        self.controller.load(path)
        dm = self.controller.diffusion_model
        self.trainer = SITrainer(dm.net.state_dict(), self.controller.state_encoder.state_dict(), gamma_type=dm.gamma_type,
                                 interpolant_type=dm.interpolant_type, beta_max=dm.d, lr=self.base_lr, weight_decay=self.trainer.wd,
                                 ema_decay=dm.ema.decay, device=self.device)
        ema_names = list(dm.net.state_dict().keys())
        self.trainer.load_ema(dict(zip(ema_names, dm.ema.shadow_params)), dm.ema.num_updates or 0)
