#!/usr/bin/env python
"""Golden vectors for the controller TRAINING step, captured from the REFERENCE ITSELF with torch autograd on CPU:
`StochasticInterpolants.get_loss` (bridge/bridge_model.py:183-258) on the reference's own `InterpolantsConditionalUnet1D`, backward,
`optim.AdamW(lr 1e-4, weight_decay 1e-6)` over net + state_encoder parameters and the EMA update (bridge_train.py:49-58, 312-334;
torch_ema is absent: its update rule is the stand-in of tools/ref_import.py), two consecutive steps.
The reference draws `step = torch.rand(B)` and `z = randn_like(x0)` inside get_loss; both are pinned here to seeded tensors the tests
regenerate.  Stored: the losses, d loss / d obs_cond, and per parameter tensor (norm, projection on a seeded random direction, first 4
values) of the gradient, of the updated parameter and of the EMA shadow -> tests/golden/g13_train.npz.
    python tools/make_golden_train.py
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

B, T = 16, 16


def train_inputs(step: int):
    g = synth.inputs_rng(500 + step)
    t = g.uniform(0, 1, B).astype(np.float32)
    t[0], t[1] = 0.0002, 0.9999                       # both clipped to [t_min, 1 - t_min]
    return dict(obs_in=cases.T(g.standard_normal((B, 781), dtype=np.float32)), vla_n=cases.T(g.uniform(-1, 1, (B, T, 10)).astype(np.float32)),
                expert_n=cases.T(g.uniform(-1, 1, (B, T, 10)).astype(np.float32)), t=cases.T(t), z=cases.T(g.standard_normal((B, T, 10), dtype=np.float32)))


def direction(name: str, shape) -> np.ndarray:
    return synth.tensor("proj." + name, tuple(shape), "train")


def summary(name: str, a: torch.Tensor) -> np.ndarray:
    v = a.detach().double().numpy()
    return np.concatenate([[np.sqrt((v * v).sum()), (v * direction(name, v.shape).astype(np.float64)).sum()], v.reshape(-1)[:4]])


def main():
    ref_import.setup()
    ref_import.no_cuda()
    from bridge.bridge_model import StochasticInterpolants  # reference
    args = dict(cases.MODEL_ARGS)
    si = StochasticInterpolants()
    si.load_model(args, "cpu")
    sd = cases.si_net_sd("")
    si.net.load_state_dict(sd)
    si.ema.shadow_params = [p.clone().detach() for p in si.net.parameters()]
    si.net.train()
    enc = torch.nn.Sequential(torch.nn.Linear(781, 256), torch.nn.GELU(), torch.nn.Linear(256, 256), torch.nn.GELU(), torch.nn.Linear(256, 256))
    enc.load_state_dict(cases.state_encoder_sd(781))                       # bridge_controller.py:42-48
    opt = torch.optim.AdamW(list(si.net.parameters()) + list(enc.parameters()), lr=1e-4, weight_decay=1e-6)
    out = {}
    orig_rand, orig_randn_like = torch.rand, torch.randn_like
    for step in (1, 2):
        inp = train_inputs(step)
        torch.rand = lambda *a, **k: inp["t"].clone()
        torch.randn_like = lambda x, *a, **k: inp["z"].clone()
        try:
            cond = enc(inp["obs_in"])
            cond.retain_grad()
            opt.zero_grad()
            loss, info = si.get_loss({"obs_cond": cond, "expert_act": inp["expert_n"], "vla_act": inp["vla_n"]}, "cpu")
            loss.backward()
        finally:
            torch.rand, torch.randn_like = orig_rand, orig_randn_like
        out[f"s{step}_loss"] = np.array([float(loss), float(info["v_loss"]), float(info["s_loss"]), float(info["b_loss"])])
        out[f"s{step}_dcond"] = cond.grad.numpy().copy()
        named = list(si.net.named_parameters()) + [("state_encoder." + k, p) for k, p in enc.named_parameters()]
        out["names"] = np.array([k for k, _ in named])
        out[f"s{step}_grad"] = np.stack([summary(k, p.grad) for k, p in named])               # [n_tensors, 6], rows in `names` order
        opt.step()
        si.ema.update()
        out[f"s{step}_param"] = np.stack([summary(k, p) for k, p in named])
        out[f"s{step}_ema"] = np.stack([summary(k, s) for (k, _), s in zip(si.net.named_parameters(), si.ema.shadow_params)])
        print(step, out[f"s{step}_loss"])
    np.savez_compressed(os.path.join(cases.GOLDEN, "g13_train.npz"), **out)
    print("wrote g13_train", len(out), "arrays")


def time_reference(batch: int = 128, iters: int = 3):
    """Wall time of the REFERENCE's own training step on this container's CPU (torch autograd + AdamW + EMA), for DESIGN §6."""
    import time
    ref_import.setup()
    ref_import.no_cuda()
    from bridge.bridge_model import StochasticInterpolants  # reference
    si = StochasticInterpolants()
    si.load_model(dict(cases.MODEL_ARGS), "cpu")
    si.net.load_state_dict(cases.si_net_sd(""))
    si.net.train()
    enc = torch.nn.Sequential(torch.nn.Linear(781, 256), torch.nn.GELU(), torch.nn.Linear(256, 256), torch.nn.GELU(), torch.nn.Linear(256, 256))
    opt = torch.optim.AdamW(list(si.net.parameters()) + list(enc.parameters()), lr=1e-4, weight_decay=1e-6)
    g = torch.Generator().manual_seed(0)
    obs, x0, x1 = torch.randn(batch, 781, generator=g), torch.randn(batch, T, 10, generator=g), torch.randn(batch, T, 10, generator=g)
    ts = []
    for i in range(iters + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = si.get_loss({"obs_cond": enc(obs), "expert_act": x1, "vla_act": x0}, "cpu")
        loss.backward()
        opt.step()
        si.ema.update()
        ts.append(time.perf_counter() - t0)
    print(f"reference training step on CPU ({torch.get_num_threads()} threads), B={batch}: {min(ts[1:]) * 1e3:.0f} ms/step "
          f"= {batch / min(ts[1:]):.0f} samples/s")


# every non-linear `interpolant_type` the reference accepts (bridge_model.py:103-181), each with one of its gamma schedules
INTERPOLANTS = [("power3", "2^0.5*t(t-1)"), ("power4", "(2t(t-1))^0.5"), ("reverse_power3", "(1-t)^2(2t)^0.5"), ("reverse_power4", "2^0.5*t(t-1)"),
                ("gaussian_encode_decode", "(2t(t-1))^0.5"), ("reverse_linear", "2^0.5*t(t-1)")]


def interpolants():
    """One get_loss + backward of the REFERENCE per non-linear interpolant (round 6): losses, d loss / d obs_cond and the per-tensor gradient
    summaries -> tests/golden/g13_train_interpolants.npz (keys `<interpolant>_loss|dcond|grad`, `names`, `<interpolant>_gamma`)."""
    ref_import.setup()
    ref_import.no_cuda()
    from bridge.bridge_model import StochasticInterpolants  # reference
    out = {}
    orig_rand, orig_randn_like = torch.rand, torch.randn_like
    inp = train_inputs(1)
    inp["t"][2], inp["t"][3] = 0.5, 0.5000001                              # both sides of the indicator of the piecewise interpolants
    for kind, gamma in INTERPOLANTS:
        args = dict(cases.MODEL_ARGS, interpolant_type=kind, gamma_type=gamma)
        si = StochasticInterpolants()
        si.load_model(args, "cpu")
        si.net.load_state_dict(cases.si_net_sd(""))
        si.net.train()
        enc = torch.nn.Sequential(torch.nn.Linear(781, 256), torch.nn.GELU(), torch.nn.Linear(256, 256), torch.nn.GELU(), torch.nn.Linear(256, 256))
        enc.load_state_dict(cases.state_encoder_sd(781))
        torch.rand = lambda *a, **k: inp["t"].clone()
        torch.randn_like = lambda x, *a, **k: inp["z"].clone()
        try:
            cond = enc(inp["obs_in"])
            cond.retain_grad()
            loss, info = si.get_loss({"obs_cond": cond, "expert_act": inp["expert_n"], "vla_act": inp["vla_n"]}, "cpu")
            loss.backward()
        finally:
            torch.rand, torch.randn_like = orig_rand, orig_randn_like
        named = list(si.net.named_parameters()) + [("state_encoder." + k, p) for k, p in enc.named_parameters()]
        out["names"] = np.array([k for k, _ in named])
        out[f"{kind}_gamma"] = np.array(gamma)
        out[f"{kind}_loss"] = np.array([float(loss), float(info["v_loss"]), float(info["s_loss"]), float(info["b_loss"])])
        out[f"{kind}_dcond"] = cond.grad.numpy().copy()
        out[f"{kind}_grad"] = np.stack([summary(k, p.grad) for k, p in named])
        print(kind, gamma, out[f"{kind}_loss"])
    out["t"] = inp["t"].numpy()
    np.savez_compressed(os.path.join(cases.GOLDEN, "g13_train_interpolants.npz"), **out)
    print("wrote g13_train_interpolants", len(out), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--interpolants":
        interpolants()
    elif len(sys.argv) > 1 and sys.argv[1] == "--time":
        time_reference(int(sys.argv[2]) if len(sys.argv) > 2 else 128)
    else:
        main()
