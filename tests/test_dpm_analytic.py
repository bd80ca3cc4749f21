"""Independent evidence for the UNPINNED DPM-Solver++(2M) leg (diffusers is absent and not version-pinned by
the reference, models/rdt_runner.py:69-76,144-160), none of it produced by the product or the oracle:

  1. per-step constants for 3 and 5 inference steps of the upstream RDT-1B scheduler config (squaredcos_cap_v2, 1000 train
     steps, linspace spacing, zero final sigma, midpoint 2M with lower_order_final), computed by hand in float64 from
     Lu et al. 2022 Alg. 2:   x_i = (sigma_i / sigma_{i-1}) x_{i-1} - alpha_i (e^{-h_i} - 1) D_i,
     D_i = (1 + 1/(2 r_i)) x0_{i-1} - (1/(2 r_i)) x0_{i-2},  r_i = h_{i-1} / h_i,  lambda = log(alpha / sigma),
     and hard-coded below;
  2. an ANALYTIC denoiser: for data ~ N(mu, s^2) the optimal x0-predictor and the probability-flow ODE solution are closed
     forms, so the sampler's output can be compared with the exact answer: the second-order multistep update must converge
     clearly faster than its first-order truncation (b1 folded into b0), which converges with order ~1.
The product's host schedule (vlatouch/dpm.py: the three scalars per step the fused HIP update kernel applies) and the
oracle's stepper (oracle/dpm_solver.py) are both held to these.
"""
import math

import pytest

import numpy as np
import torch

from vlatouch import dpm
from oracle import dpm_solver

# (sigma_i/sigma_{i-1}, coefficient of x0_k, coefficient of x0_{k-1}) — hand-computed, float64
HAND = {
    3: ([999, 666, 333],
        [(0.868331414, 0.495941640, 0.000000000),
         (0.584986157, 0.604447774, -0.033212238),
         (0.000000000, 1.000000000, 0.000000000)]),
    5: ([999, 799, 599, 400, 200],
        [(0.951816363, 0.306621663, 0.000000000),
         (0.853006337, 0.336919265, -0.014720463),
         (0.732830379, 0.523556682, -0.147641132),
         (0.537397132, 0.836740386, -0.321161157),
         (0.000000000, 1.000000000, 0.000000000)]),
}


def test_coefficients_match_hand_computed_constants():
    for n, (ts, rows) in HAND.items():
        pts, coef = dpm.schedule(1000, "squaredcos_cap_v2", n)
        assert pts == ts
        np.testing.assert_allclose(coef[:, :3], np.array(rows), rtol=0, atol=3e-6)
        s = dpm_solver.DPMSolverPP2M(1000, "squaredcos_cap_v2", "sample")
        s.set_timesteps(n)
        assert s.timesteps == ts
        oc = np.array([[c["a"], c["b0"], c["b1"]] for c in s.coefficients()])
        np.testing.assert_allclose(oc, np.array(rows), rtol=0, atol=3e-6)


def _alphas_cumprod(N=1000):
    bar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    ac, p = [], 1.0
    for i in range(N):
        p *= 1.0 - min(1 - bar((i + 1) / N) / bar(i / N), 0.999)
        ac.append(p)
    return np.array(ac)


MU, S = 0.7, 0.5


def _gaussian_problem(ts):
    ac = _alphas_cumprod()
    a_t, s_t = np.sqrt(ac[ts]), np.sqrt(1 - ac[ts])
    var = lambda k: a_t[k] ** 2 * S * S + s_t[k] ** 2
    xT = a_t[0] * MU + math.sqrt(var(0)) * np.linspace(-2, 2, 9)
    x0_pred = lambda k, x: MU + (a_t[k] * S * S) / var(k) * (x - a_t[k] * MU)             # E[x0 | x_t]
    exact_final = MU + S / math.sqrt(var(0)) * (xT - a_t[0] * MU)                          # PF-ODE solution at t = 0
    return xT, x0_pred, exact_final


def _run_product_schedule(n, second_order=True):
    ts, coef = dpm.schedule(1000, "squaredcos_cap_v2", n)
    coef = coef.astype(np.float64)
    xT, x0_pred, exact = _gaussian_problem(ts)
    x, prev = xT.copy(), None
    for k in range(n):
        x0 = x0_pred(k, x)
        a, b0, b1 = coef[k, :3]
        if not second_order:
            b0, b1 = b0 + b1, 0.0
        x = a * x + b0 * x0 + (b1 * prev if prev is not None else 0.0)
        prev = x0
    return float(np.abs(x - exact).max())


def test_analytic_gaussian_second_order_convergence():
    ns = [10, 20, 40, 80, 160]
    e2 = [_run_product_schedule(n) for n in ns]
    e1 = [_run_product_schedule(n, second_order=False) for n in ns]
    slope = lambda e: -np.polyfit(np.log(ns), np.log(e), 1)[0]
    print("2M errors", e2, "order", slope(e2), "| first-order errors", e1, "order", slope(e1))
    assert 0.8 < slope(e1) < 1.2                       # the truncated scheme is first order
    assert slope(e2) > 1.7                             # the multistep correction buys (at least) one more order
    assert all(b < a / 8 for a, b in zip(e1[1:], e2[1:]))
    assert e2[-1] < 1e-4                               # and it converges to the analytic marginal


def test_oracle_stepper_tracks_the_same_trajectory():
    n = 20
    s = dpm_solver.DPMSolverPP2M(1000, "squaredcos_cap_v2", "sample")
    s.set_timesteps(n)
    xT, x0_pred, exact = _gaussian_problem(np.array(s.timesteps))
    x = torch.from_numpy(xT).float()
    for k in range(n):
        x = s.step(torch.from_numpy(x0_pred(k, x.double().numpy())).float(), x)
    assert float(np.abs(x.numpy() - exact).max()) < 5e-3
    assert abs(float(np.abs(x.numpy() - exact).max()) - _run_product_schedule(n)) < 1e-4


def test_epsilon_prediction_is_the_same_update():
    """prediction_type 'epsilon': x0 = (x - sigma_s eps) / alpha_s, then the same three-scalar update."""
    n = 5
    ts, coef = dpm.schedule(1000, "squaredcos_cap_v2", n)
    ac = _alphas_cumprod()
    np.testing.assert_allclose(coef[:, 3], np.sqrt(ac[ts]), atol=2e-6)
    np.testing.assert_allclose(coef[:, 4], np.sqrt(1 - ac[ts]), atol=2e-6)


def test_schedule_against_diffusers_when_installed():
    """Where diffusers IS importable (a maintainer's machine, not this image): the product's folded update x <- a x + b0 x0_k + b1 x0_{k-1}
    must reproduce `DPMSolverMultistepScheduler.step` on random tensors, with the scheduler constructed as models/rdt_runner.py:69-76 does."""
    diffusers = pytest.importorskip("diffusers")
    import torch
    from vlatouch import dpm
    for n in (3, 5, 50):
        sch = diffusers.DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2", prediction_type="sample")
        sch.set_timesteps(n)
        ts, coef = dpm.schedule(1000, "squaredcos_cap_v2", n)
        assert [int(t) for t in sch.timesteps] == [int(t) for t in ts]
        g = torch.Generator().manual_seed(n)
        x = torch.randn(2, 64, 128, generator=g, dtype=torch.float64)
        mine, prev = x.clone(), None
        for k, t in enumerate(sch.timesteps):
            x0 = torch.randn(2, 64, 128, generator=g, dtype=torch.float64)
            x = sch.step(x0, t, x).prev_sample
            a, b0, b1 = (float(c) for c in coef[k][:3])
            mine = a * mine + b0 * x0 + (b1 * prev if prev is not None else 0.0)
            prev = x0
            assert float((mine - x).abs().max()) < 5e-5 * max(1.0, float(x.abs().max())), (n, k)
