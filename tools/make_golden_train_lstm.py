#!/usr/bin/env python
"""Golden vectors for the LSTM residual head's TRAINING step, captured from the REFERENCE ITSELF with torch autograd on
CPU: `TactileLSTMController.get_loss` (residual_controller/lstm_step_controller.py:176-211, 321-337) on the reference's own modules,
backward, `optim.AdamW(lr 1e-4, weight_decay 1e-6)` over `trainable_modules` + `CosineAnnealingLR(T_max 100000, eta_min lr/10)`
(lstm_train.py:26-33, 129-133), two consecutive steps.  The controller is put in `.eval()` so that the two dropouts (nn.LSTM inter-layer
p=0.1, output head p=0.1) are inactive — their masks come from torch's generator and cannot be pinned; the masked arithmetic is covered
in tests/test_gpu_train.py against a torch restatement with injected masks.  obs_cond is produced inside the graph by `obs_encoder`
(as `encode_observation` does, :139-158) from a seeded [cls_cam1 | cls_cam2 | state] row.
Stored per step: loss, prediction, d loss / d obs_cond, and per parameter tensor (norm, projection on a seeded direction, first 4 values)
of the gradient and of the updated parameter -> tests/golden/g14_train_lstm.npz.
    python tools/make_golden_train_lstm.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from tests import cases  # noqa: E402
import ref_import  # noqa: E402
from vlatouch import synth  # noqa: E402
from make_golden_train import summary  # noqa: E402

B, T = 16, 16
MODS = ("obs_encoder", "force_encoder", "lstm", "output_head")


def lstm_train_inputs(step: int):
    g = synth.inputs_rng(700 + step)
    return dict(obs_in=cases.T(g.standard_normal((B, 778), dtype=np.float32)), vla_n=cases.T(g.uniform(-1, 1, (B, T, 10)).astype(np.float32)),
                forces=cases.T(g.standard_normal((B, T, 3), dtype=np.float32)), expert_n=cases.T(g.uniform(-1, 1, (B, T, 10)).astype(np.float32)))


def main():
    ref_import.setup()
    ref_import.no_cuda()

    class _Enc(torch.nn.Module):                     # the image encoder is not on this path; only its hidden size is read
        class config:
            hidden_size = 384
    ref_import.patch_dinov2(lambda name: _Enc())
    from lstm_step_controller import TactileLSTMController  # reference
    lc = TactileLSTMController(state_dim=10, hidden_dim=256, num_layers=2, dropout=0.1, image_model_path="facebook/dinov2-small", device="cpu",
                               force_dim=3)
    mods = cases.lstm_mods()
    for m in MODS:
        getattr(lc, m).load_state_dict(mods[m])
    lc.eval()
    params = [p for m in lc.trainable_modules for p in m.parameters()]
    named = [(f"{m}.{k}", p) for m in MODS for k, p in getattr(lc, m).named_parameters()]
    assert len(named) == len(params)
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-6)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100000, eta_min=1e-4 / 10)
    out = {"names": np.array([k for k, _ in named])}
    for step in (1, 2):
        inp = lstm_train_inputs(step)
        cond = lc.obs_encoder(inp["obs_in"])
        cond.retain_grad()
        opt.zero_grad()
        bd = {"obs_cond": cond, "vla_act": inp["vla_n"], "forces": inp["forces"], "expert_act": inp["expert_n"]}
        pred = lc.forward(bd)
        loss = lc.get_loss(bd)
        loss.backward()
        out[f"s{step}_loss"] = np.array([float(loss)])
        out[f"s{step}_pred"] = pred.detach().numpy().copy()
        out[f"s{step}_dcond"] = cond.grad.numpy().copy()
        out[f"s{step}_grad"] = np.stack([summary(k, p.grad) for k, p in named])
        opt.step()
        sched.step()
        out[f"s{step}_param"] = np.stack([summary(k, p) for k, p in named])
        print(step, float(loss), "lr next", sched.get_last_lr())
    np.savez_compressed(os.path.join(cases.GOLDEN, "g14_train_lstm.npz"), **out)
    print("wrote g14_train_lstm", len(out), "arrays")


if __name__ == "__main__":
    main()
