"""Oracle: DPM-Solver++(2M) multistep scheduler (test infrastructure).  PARITY UNPINNED.

The reference samples RDT with diffusers' `DPMSolverMultistepScheduler`
(/root/reference/VLA/models/rdt_runner.py:69-76,144,158).  diffusers is absent from
/root/reference and from this image and is not version-pinned by the reference (upstream
RDT-1B pins diffusers==0.27.2).  This restates the published algorithm (Lu et al. 2022,
"DPM-Solver++", Alg. 2, multistep 2M, data-prediction form) with diffusers' defaults:
solver_order=2, algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
timestep_spacing="linspace", final_sigmas_type="zero", no thresholding, no Karras sigmas.
All scheduler math is fp32 on the up-cast model output; the caller casts back (rdt_runner.py:160).
"""
from __future__ import annotations

import math
from typing import List

import numpy as np
import torch


def make_betas(num_train_timesteps: int, beta_schedule: str, beta_start: float = 1e-4, beta_end: float = 0.02) -> np.ndarray:
    if beta_schedule == "squaredcos_cap_v2":
        f = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        n = num_train_timesteps
        return np.array([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)], dtype=np.float32)
    if beta_schedule == "linear":
        return np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
    if beta_schedule == "scaled_linear":
        return np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
    raise NotImplementedError(beta_schedule)


class DPMSolverPP2M:
    def __init__(self, num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2",
                 prediction_type: str = "sample"):
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        betas = torch.from_numpy(make_betas(num_train_timesteps, beta_schedule))
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)

    def set_timesteps(self, n: int):
        N = self.num_train_timesteps
        ts = np.linspace(0, N - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        ac = self.alphas_cumprod.numpy()
        sig = ((1 - ac) / ac) ** 0.5
        sig = np.interp(ts, np.arange(N), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = [int(t) for t in ts]
        self.prev_x0 = None
        self.idx = 0
        self.lower_order_nums = 0

    @staticmethod
    def _alpha_sigma(sigma: torch.Tensor):
        alpha = 1.0 / torch.sqrt(sigma ** 2 + 1.0)
        return alpha, sigma * alpha

    def coefficients(self) -> List[dict]:
        """Per-step scalars (a, b0, b1) with x <- a*x + b0*x0_k + b1*x0_{k-1} for `sample`
        prediction — what the HIP path folds into one element-wise kernel per step."""
        out = []
        n = len(self.timesteps)
        for i in range(n):
            at, st = self._alpha_sigma(self.sigmas[i + 1])
            a0, s0 = self._alpha_sigma(self.sigmas[i])
            lam_t, lam_0 = torch.log(at) - torch.log(st), torch.log(a0) - torch.log(s0)
            h = lam_t - lam_0
            first = (i == 0) or (i == n - 1)          # lower_order_final with < 15 steps / zero final sigma
            e = at * (torch.exp(-h) - 1.0)
            if first:
                out.append(dict(a=float(st / s0), b0=float(-e), b1=0.0, alpha_s=float(a0), sigma_s=float(s0)))
            else:
                a1, s1 = self._alpha_sigma(self.sigmas[i - 1])
                lam_1 = torch.log(a1) - torch.log(s1)
                r0 = (lam_0 - lam_1) / h
                c = 0.5 * e / r0
                out.append(dict(a=float(st / s0), b0=float(-e - c), b1=float(c), alpha_s=float(a0), sigma_s=float(s0)))
        return out

    def step(self, model_output: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        i, n = self.idx, len(self.timesteps)
        x = sample.float()
        m = model_output.float()
        if self.prediction_type == "sample":
            x0 = m
        elif self.prediction_type == "epsilon":
            a_s, s_s = self._alpha_sigma(self.sigmas[i])
            x0 = (x - s_s * m) / a_s
        else:
            raise ValueError(self.prediction_type)
        at, st = self._alpha_sigma(self.sigmas[i + 1])
        a0, s0 = self._alpha_sigma(self.sigmas[i])
        lam_t, lam_0 = torch.log(at) - torch.log(st), torch.log(a0) - torch.log(s0)
        h = lam_t - lam_0
        lower_final = (i == n - 1)
        if self.lower_order_nums < 1 or lower_final:
            x_new = (st / s0) * x - (at * (torch.exp(-h) - 1.0)) * x0
        else:
            a1, s1 = self._alpha_sigma(self.sigmas[i - 1])
            lam_1 = torch.log(a1) - torch.log(s1)
            r0 = (lam_0 - lam_1) / h
            D0, D1 = x0, (1.0 / r0) * (x0 - self.prev_x0)
            e = at * (torch.exp(-h) - 1.0)
            x_new = (st / s0) * x - e * D0 - 0.5 * e * D1
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.prev_x0 = x0
        self.idx += 1
        return x_new
