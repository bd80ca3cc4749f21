"""Mirror of the reference's `LSTMControllerTrainer` (VLA/residual_controller/lstm_train.py:19) on the device training step.

Same constructor, `_prepare_batch`, `train`, `_train_epoch`, `_evaluate`, `_save_checkpoint` (:20-40, :56-84, :86-106, :108-140,
:142-160).  The optimiser and the backward pass are `vlatouch.train.LstmTrainer` (HIP kernels through the C ABI): AdamW over the four
`trainable_modules` (:26-30), cosine LR (:31-33), back-propagation through the T ticks of both LSTM layers.  `_prepare_batch` returns
`obs_in` ([cls_cam1 | cls_cam2 | state], the obs_encoder's input) next to the other entries because the obs_encoder is trained through the
loss; TensorBoard / file logging are replaced by `self.history`.  In `controller.train()` mode the two dropouts draw keep-masks on the
device; `_evaluate` runs without them, as `controller.eval()` does in the reference.
"""
from __future__ import annotations

import os
from typing import Dict

import torch

from vlatouch import _lib as L
from vlatouch.engine import concat_obs
from vlatouch.train import LstmTrainer

from .controller_dataset import normalize_actions


class LSTMControllerTrainer:
    def __init__(self, controller, data_module, learning_rate=1e-4, weight_decay=1e-6, checkpoint_dir='checkpoint/lstm_controller', device='cuda'):
        self.controller = controller
        self.device = device
        self.checkpoint_dir = checkpoint_dir
        os.makedirs(checkpoint_dir, exist_ok=True)
        self.weight_decay = weight_decay
        self.controller.stats = {k: torch.as_tensor(v, dtype=torch.float32).to(device) for k, v in data_module.stats.items()}
        self.data_module = data_module
        self.trainer = self._make_trainer(learning_rate)
        self.history = []
        self._dirty = False

    def _make_trainer(self, lr):
        c = self.controller
        mods = {"obs_encoder": c.obs_encoder.state_dict(), "force_encoder": c.force_encoder.state_dict(), "lstm": c.lstm.state_dict(),
                "output_head": c.output_head.state_dict()}
        return LstmTrainer(mods, lr=lr, weight_decay=self.weight_decay, device=self.device)

    def _prepare_batch(self, batch) -> Dict[str, torch.Tensor]:
        cf = 2
        current_state = batch['states'][:, cf - 1]
        forces = batch['forces'][:, cf - 1:-1]
        vla_n = normalize_actions(batch['vla_actions'], self.controller.stats, 'vla')
        expert_n = normalize_actions(batch['expert_actions'], self.controller.stats, 'expert')
        f1, f2 = self.controller.encode_images(batch['images_cam1'][:, -1], batch['images_cam2'][:, -1])
        obs_in = concat_obs(f1, f2, torch.as_tensor(current_state), None, self.trainer.obs.kin, L.F32, torch.device(self.device))
        return {'obs_in': obs_in, 'expert_act': expert_n, 'vla_act': vla_n, 'forces': forces}

    def train_step(self, batch, masks="draw"):
        """One iteration of the reference's loop body (:116-133)."""
        bd = self._prepare_batch(batch)
        loss = self.trainer.train_step(bd['obs_in'], bd['vla_act'], bd['forces'], bd['expert_act'], masks=masks)
        self._dirty = True
        return loss

    def eval_step(self, batch):
        bd = self._prepare_batch(batch)
        return self.trainer.get_loss(bd['obs_in'], bd['vla_act'], bd['forces'], bd['expert_act'], masks=None, backward=False)[0]

    def sync_controller(self) -> None:
        if not self._dirty:
            return
        for name, sd in self.trainer.modules_state_dict().items():
            getattr(self.controller, name).load_state_dict(sd)
        self._dirty = False

    def _train_epoch(self, train_loader, epoch, global_step, log_interval):
        running, n = 0.0, 0
        for batch in train_loader:
            loss = self.train_step(batch)
            running, n = running + loss, n + 1
            if global_step % log_interval == 0:
                self.history.append({"step": global_step, "epoch": epoch + 1, "loss": loss, "lr": self.trainer.lr})
            global_step += 1
        return running / max(n, 1), global_step

    def _evaluate(self, val_loader, epoch):
        self.controller.eval()
        running, n = 0.0, 0
        for batch in val_loader:
            running, n = running + self.eval_step(batch), n + 1
        avg = running / max(n, 1)
        self.history.append({"epoch": epoch, "val_loss": avg})
        return avg

    def train(self, data_module, num_epochs=500, save_interval=50, log_interval=100):
        train_loader, val_loader = data_module.train_dataloader(), data_module.val_dataloader()
        global_step, best = 0, float('inf')
        for epoch in range(num_epochs):
            self.controller.train()
            _, global_step = self._train_epoch(train_loader, epoch, global_step, log_interval)
            if (epoch + 1) % 5 == 0:
                val = self._evaluate(val_loader, epoch + 1)
                if val < best:
                    best = val
                    self._save_checkpoint("best_model")
                if (epoch + 1) % save_interval == 0:
                    self._save_checkpoint(f"epoch_{epoch + 1}")
        return best

    def _save_checkpoint(self, name):
        path = os.path.join(self.checkpoint_dir, name)
        os.makedirs(path, exist_ok=True)
        self.sync_controller()
        self.controller.save(path)

    def load_checkpoint(self, path):
        self.controller.load(path)
        self.trainer = self._make_trainer(self.trainer.base_lr)
