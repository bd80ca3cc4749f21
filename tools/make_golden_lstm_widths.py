#!/usr/bin/env python
"""Golden vectors of the LSTM residual head at widths other than the default (`lstm_train.py --hidden_dim`): the REFERENCE's own
TactileLSTMController (imported read-only in this container) at hidden_dim 128 (2 layers) and 384 (3 layers), synthetic weights from tests/cases.py,
only the OUTPUTS stored:

    python tools/make_golden_lstm_widths.py      # writes tests/golden/g6_lstm_h128l2.npz, g6_lstm_h384l3.npz
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

torch.set_grad_enabled(False)
ref_import.setup()
ref_import.no_cuda()


def build_dino(name: str):
    from transformers import Dinov2Config, Dinov2Model
    c = synth.DINOV2_CONFIGS["small"]
    cfg = Dinov2Config(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"], image_size=518, patch_size=14, mlp_ratio=4,
                       qkv_bias=True, layerscale_value=1.0, hidden_act="gelu", layer_norm_eps=1e-6, use_swiglu_ffn=False)
    m = Dinov2Model(cfg).eval()
    m.load_state_dict({k: cases.T(synth.tensor("dinov2-small." + k, tuple(v.shape), "")).to(v.dtype) for k, v in m.state_dict().items()})
    return m


ref_import.patch_dinov2(build_dino)
from lstm_step_controller import TactileLSTMController  # noqa: E402  (reference)
import oracle.normalize as onorm  # noqa: E402

for hidden, layers in ((128, 2), (384, 3)):
    lc = TactileLSTMController(state_dim=10, hidden_dim=hidden, num_layers=layers, dropout=0.1, image_model_path="facebook/dinov2-small", device="cpu", force_dim=3)
    mods = cases.lstm_mods(384, hidden=hidden, layers=layers)
    for mname in ("obs_encoder", "force_encoder", "lstm", "output_head"):
        getattr(lc, mname).load_state_dict(mods[mname])
    lc.stats = cases.stats("nontrivial")
    lc.eval()
    li = cases.lstm_inputs(3, 16, hidden=hidden)
    vn = onorm.normalize_actions(li["vla"], lc.stats, "vla")
    fwd = lc.forward({"vla_act": vn, "obs_cond": li["obs_cond"], "forces": li["forces"]})
    seq = lc.predict_sequence(li["obs_cond"], li["vla"], li["forces"])
    pi = cases.predict_inputs(2, 16, 224)
    obs_l = lc.encode_observation(pi["state"], pi["cam1"], pi["cam2"])
    out = dict(forward=fwd.numpy(), predict_sequence=seq.numpy(), obs_cond=obs_l.numpy(), h=lc.hidden_state.numpy(), c=lc.cell_state.numpy())
    np.savez_compressed(os.path.join(cases.GOLDEN, f"g6_lstm_h{hidden}l{layers}.npz"), **out)
    print(hidden, layers, {k: v.shape for k, v in out.items()})
