#!/usr/bin/env python
"""Time the device training steps at the reference's default batch size (bridge_train.py:698: 128; horizon 16).

    python tools/train_bench.py [--batch 128] [--steps 20] [--json profiles/rNN_bench_train_n1.json]

One JSON line per head in bench.py's vocabulary: `value` = training samples/s of the hipGraph-replayed step (interpolant controller) or the
step of either head, inputs resident on the device; `roofline` = the dominant kernel class of the step, the exact-fp32 MFMA
GEMMs (`gemm_f32r_kernel`, `gemm_kernel<float,float,float,...>`): algorithmic 2·M·N·K flops of its launches in ONE eager step ÷ the HIP-event time of those
launches (vt_prof class 5), against the fp32 matrix peak (256 CUs x 256 flop/clk x 2.4 GHz = 157.3 TFLOP/s, MI355X_MICROARCH.md).
No `cpu_baseline`: the CPU side of this row is the reference's own torch autograd step, which cannot travel to the GPU box; its time in the
build container (`python tools/make_golden_train.py --time 128`: 882 ms/step on 8 threads) is recorded in DESIGN §6.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vla-touch_amd"))
from tests import cases  # noqa: E402

FP32_MFMA_PEAK = 157.3


def timeit(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def gemm_leg(step_fn):
    """One eager step with HIP events around every launch of the register-staged GEMM -> (ms, flops, launches)."""
    from vlatouch import _lib as L
    lib = L.lib()
    lib.vt_prof_enable(5)
    step_fn()
    torch.cuda.synchronize()
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_long()
    L.check(lib.vt_prof_collect(C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "vt_prof_collect")
    lib.vt_prof_enable(0)
    return ms.value, fl.value, n.value


def line(metric, B, steps, ms, hipgraph, nparam, leg, what):
    gms, gfl, gn = leg
    ach = gfl / (gms * 1e-3) / 1e12 if gms > 0 else None
    return {"metric": metric, "value": round(B / ms * 1e3, 1), "unit": "samples/s", "n_gpus": 1, "steps": steps, "warmup": 3,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "batch_per_gpu": B, "horizon": 16, "trained_parameters": nparam, "hipgraph": hipgraph},
            "roofline": {"bound": "mfma", "kernel": "gemm_f32r_kernel / gemm_kernel<float,float,float,*> (exact fp32 v_mfma_f32_16x16x4_f32: LDS-DMA ring for small grids, register-staged otherwise)",
                         "achieved": None if ach is None else round(ach, 2), "peak": FP32_MFMA_PEAK, "unit": "TFLOP/s",
                         "frac": None if ach is None else round(ach / FP32_MFMA_PEAK, 4), "traffic": None,
                         "launches_per_step": gn, "kernel_ms_per_step": round(gms, 3), "algorithmic_gflop_per_step": round(gfl / 1e9, 1)}}


def bench_si(B, steps, dev="cuda:0"):
    from vlatouch.train import SITrainer
    T = 16
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    si = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), device=dev)
    nparam = sum(p.numel() for _, p, _ in si._all_params())
    obs, x0, x1, t, z = r(B, 781), r(B, T, 10), r(B, T, 10), torch.rand(B, generator=g).to(dev), r(B, T, 10)
    step = lambda: si.train_step(obs, x0, x1, t, z, sync=False)
    for _ in range(3):
        step()
    eager = timeit(step, steps)
    leg = gemm_leg(step)
    si.capture(B)
    gstep = lambda: si.replay(obs, x0, x1, t, z)
    for _ in range(3):
        gstep()
    ms = timeit(gstep, steps)
    print(f"[train_bench] interpolant controller: {nparam / 1e6:.2f} M parameters, B={B}: eager {eager:.2f} ms/step, hipGraph replay {ms:.2f} ms/step",
          file=sys.stderr)
    return line("interpolant-controller training samples/sec (get_loss + backward + AdamW + EMA)", B, steps, ms, True, nparam, leg,
                "train_si (SURVEY 8f-4): StochasticInterpolants.get_loss over v/s/b U-Nets + observation MLP, backward, AdamW(1e-4, wd 1e-6), "
                "EMA 0.75; one step = one batch; replayed from one hipGraph, eager %.2f ms" % eager)


def bench_lstm(B, steps, dev="cuda:0"):
    from vlatouch.train import LstmTrainer
    T = 16
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    lt = LstmTrainer(cases.lstm_mods(), device=dev)
    nl = sum(p.numel() for _, p, _ in lt._all())
    o2, f2, x0, x1 = r(B, 778), r(B, T, 3), r(B, T, 10), r(B, T, 10)
    lstep = lambda: lt.train_step(o2, x0, f2, x1, masks="draw")
    for _ in range(3):
        lstep()
    leager = timeit(lstep, steps)
    lleg = gemm_leg(lstep)
    lt.capture(B)
    lg = lambda: lt.replay(o2, x0, f2, x1)
    for _ in range(3):
        lg()
    lms = timeit(lg, steps)
    print(f"[train_bench] LSTM head: {nl / 1e6:.2f} M parameters, B={B}: eager {leager:.2f} ms/step, hipGraph replay {lms:.2f} ms/step", file=sys.stderr)
    return line("LSTM-head training samples/sec (get_loss + BPTT + AdamW)", B, steps, lms, True, nl, lleg,
                "train_lstm (SURVEY 8f-4): TactileLSTMController.get_loss (force MLP, 2-layer LSTM over 16 ticks, residual head, dropout masks "
                "drawn on the device), back-propagation through time, AdamW + cosine LR; replayed from one hipGraph, eager %.2f ms" % leager)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    out = [bench_si(a.batch, a.steps), bench_lstm(a.batch, a.steps)]
    for o in out:
        print(json.dumps(o))
    if a.json:
        with open(a.json, "w") as f:
            for o in out:
                f.write(json.dumps(o) + "\n")


if __name__ == "__main__":
    main()
