"""DPM-Solver++(2M) coefficients for the RDT sampling loop (host side, a handful of scalars per call).

The reference steps the loop with diffusers' `DPMSolverMultistepScheduler` (models/rdt_runner.py:69-76,144,158),
a third-party package that is not vendored or version-pinned by the reference ("parity unpinned").
This is the product's own statement of the published algorithm (Lu et al. 2022, multistep second-order,
data-prediction form) with diffusers' defaults as the reference constructs it: solver_order 2, midpoint,
lower_order_final, `linspace` timestep spacing, zero final sigma, no thresholding / Karras sigmas.  The update of
step k is folded into three scalars,   x <- a_k x + b0_k x0_k + b1_k x0_{k-1},   applied by one fused kernel.

bf16 note: the kernel evaluates the update in fp32 from the bf16 x0 and rounds the state to bf16 ONCE per step (the reference's
`.to(dtype)`, rdt_runner.py:160).  That is what diffusers >= 0.28 does (it up-casts sample and model output to fp32 inside `step`);
diffusers 0.27.2 — upstream RDT-1B's pin — keeps the intermediate products in bf16 (0-dim fp32 sigmas do not promote) and can differ
by a few bf16 ulps per step.  Independent checks of the coefficients: tests/test_dpm_analytic.py.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np


def _betas(num_train_timesteps: int, beta_schedule: str, beta_start: float = 1e-4, beta_end: float = 0.02) -> np.ndarray:
    if beta_schedule == "squaredcos_cap_v2":
        bar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        n = num_train_timesteps
        return np.array([min(1 - bar((i + 1) / n) / bar(i / n), 0.999) for i in range(n)], dtype=np.float32)
    if beta_schedule == "linear":
        return np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
    if beta_schedule == "scaled_linear":
        return np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
    raise NotImplementedError(f"{beta_schedule} is not implemented")


def schedule(num_train_timesteps: int, beta_schedule: str, num_inference_steps: int) -> Tuple[List[int], np.ndarray]:
    """-> (timesteps, coef[n,5] = a, b0, b1, alpha_s, sigma_s) in fp32 arithmetic."""
    f = np.float32
    ac = np.cumprod((1.0 - _betas(num_train_timesteps, beta_schedule)).astype(np.float32), dtype=np.float32)
    n, N = num_inference_steps, num_train_timesteps
    ts = np.linspace(0, N - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
    sig_all = ((1 - ac) / ac) ** 0.5
    sig = np.concatenate([np.interp(ts, np.arange(N), sig_all), [0.0]]).astype(np.float32)

    def alpha_sigma(s):
        a = f(1.0) / np.sqrt(s * s + f(1.0), dtype=np.float32)
        return a, s * a

    with np.errstate(divide="ignore"):
        lam = []
        for s in sig:
            a, sg = alpha_sigma(f(s))
            lam.append(np.log(a) - np.log(sg))
    coef = np.zeros((n, 5), dtype=np.float32)
    for i in range(n):
        at, st = alpha_sigma(f(sig[i + 1]))
        a0, s0 = alpha_sigma(f(sig[i]))
        h = lam[i + 1] - lam[i]
        e = at * (np.exp(-h, dtype=np.float32) - f(1.0))
        first_order = (i == 0) or (i == n - 1)          # warm-up step and lower_order_final (zero final sigma / < 15 steps)
        if first_order:
            coef[i] = (st / s0, -e, 0.0, a0, s0)
        else:
            r0 = (lam[i] - lam[i - 1]) / h
            c = f(0.5) * e / r0
            coef[i] = (st / s0, -e - c, c, a0, s0)
    return [int(t) for t in ts], coef
