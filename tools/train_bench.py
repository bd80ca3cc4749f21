#!/usr/bin/env python
"""Time the device training steps at the reference's default batch size (bridge_train.py:698: 128; horizon 16):
eager (one C-ABI call per primitive from Python) and, where it captures, replayed from a hipGraph.
    python tools/train_bench.py [--batch 128] [--steps 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vla-touch_amd"))
from tests import cases  # noqa: E402


def timeit(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from vlatouch.train import SITrainer, LstmTrainer
    dev, B, T = "cuda:0", a.batch, 16
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    si = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), device=dev)
    nparam = sum(p.numel() for _, p, _ in si._all_params())
    obs, x0, x1, t, z = r(B, 781), r(B, T, 10), r(B, T, 10), torch.rand(B, generator=g).to(dev), r(B, T, 10)
    step = lambda: si.train_step(obs, x0, x1, t, z, sync=False)
    for _ in range(3):
        step()
    ms = timeit(step, a.steps)
    print(f"interpolant controller: {nparam / 1e6:.2f} M trained parameters, B={B}: eager {ms:.2f} ms/step = {B / ms * 1e3:.0f} samples/s")
    try:
        si.capture(B)
        gstep = lambda: si.replay(obs, x0, x1, t, z)
        for _ in range(3):
            gstep()
        ms = timeit(gstep, a.steps)
        print(f"interpolant controller: B={B}: hipGraph replay {ms:.2f} ms/step = {B / ms * 1e3:.0f} samples/s")
    except Exception as e:  # noqa: BLE001
        print("graph capture failed:", repr(e)[:300])
    lt = LstmTrainer(cases.lstm_mods(), device=dev)
    nl = sum(p.numel() for _, p, _ in lt._all())
    o2, f2 = r(B, 778), r(B, T, 3)
    lstep = lambda: lt.train_step(o2, x0, f2, x1, masks="draw")
    for _ in range(3):
        lstep()
    ms = timeit(lstep, a.steps)
    print(f"LSTM head: {nl / 1e6:.2f} M trained parameters, B={B}: eager {ms:.2f} ms/step = {B / ms * 1e3:.0f} samples/s")


if __name__ == "__main__":
    main()
