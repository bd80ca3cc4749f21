"""Oracle: stochastic-interpolant sampler (test infrastructure; see oracle/__init__.py).

Restates /root/reference/VLA/residual_controller/bridge/bridge_model.py: schedules :59-101,
`sample` :259-279 (EMA weights, sde_type dispatch) and `sde_vs` :334-387, with the Gaussian
noise injected (`z[k]` = the reference's k-th `torch.randn_like` draw) instead of RNG-matched.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch

T_MIN = 1e-3            # bridge_model.py:45
GAMMA_INV_MAX = 200.0   # bridge_model.py:46


def epsilon(t: torch.Tensor, kind: str) -> torch.Tensor:
    """bridge_model.py:59-71."""
    if kind == "t(t-1)":
        return t * (1 - t)
    if kind == "1-t":
        return (1 - t) * 1.0
    if kind == "1-sqrt(t)":
        return 1 - torch.sqrt(t)
    if kind == "1-t^2":
        return 1 - torch.pow(t, 2)
    if kind == "0":
        return t * 0.0
    raise NotImplementedError(kind)


def gamma(t: torch.Tensor, kind: str) -> torch.Tensor:
    """bridge_model.py:73-81 (note the literal 1.4142)."""
    if kind == "(2t(t-1))^0.5":
        return 1.4142 * torch.sqrt(t * (1 - t))
    if kind == "2^0.5*t(t-1)":
        return 1.4142 * t * (1 - t)
    if kind == "(1-t)^2(2t)^0.5":
        return 1.4142 * torch.pow((1 - t), 2.0) * torch.sqrt(t)
    raise NotImplementedError(kind)


def gamma_der(t: torch.Tensor, kind: str) -> torch.Tensor:
    """bridge_model.py:83-91."""
    if kind == "(2t(t-1))^0.5":
        return (1 - 2 * t) / torch.sqrt(2 * (t - torch.pow(t, 2)) + 1e-4)
    if kind == "2^0.5*t(t-1)":
        return 1.4142 * (1 - 2 * t)
    if kind == "(1-t)^2(2t)^0.5":
        return 1.4142 * (2 * (t - 1) * torch.sqrt(t) + torch.pow((1 - t), 2.0) / (2.0 * torch.sqrt(t + 1e-4)))
    raise NotImplementedError(kind)


def gamma_inv(t: torch.Tensor, kind: str) -> torch.Tensor:
    """bridge_model.py:93-101."""
    if kind == "(2t(t-1))^0.5":
        return torch.clamp(1 / (1.4142 * torch.sqrt(t * (1 - t) + 1e-4)), 0.0, GAMMA_INV_MAX)
    if kind == "2^0.5*t(t-1)":
        return torch.clamp(1 / (1.4142 * t * (1 - t) + 1e-4), 0.0, GAMMA_INV_MAX)
    if kind == "(1-t)^2(2t)^0.5":
        return torch.clamp(1 / (1.4142 * torch.pow((1 - t), 2.0) * torch.sqrt(t) + 1e-4), 0.0, GAMMA_INV_MAX)
    raise NotImplementedError(kind)


def sde_bs(b_net: Callable, s_net: Callable, x_initial: torch.Tensor, cond: torch.Tensor,
           noise: torch.Tensor, diffuse_step: int = 10, beta_max: float = 0.03,
           gamma_type: str = "2^0.5*t(t-1)", epsilon_type: str = "1-t", score_weight: float = 1.0,
           direction: str = "forward") -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """Drift-score SDE (bridge_model.py:281-332): forward x += (b + w*eps*s*gamma_inv)*dt + dt*sqrt(2 eps)*d*z; backward (:297-301,
    :320-323): nets / gamma_inv / eps at 1 - t, x -= (b - w*eps*s)*dt."""
    delta_t = float(1.0 / diffuse_step)
    n_steps = int(1.0 / delta_t)
    n = x_initial.shape[0]
    xs = [x_initial]
    for k in range(1, n_steps + 1):
        x = xs[-1]
        t = torch.clip(torch.full((n,), k / n_steps).float(), T_MIN, 1.0 - T_MIN)
        tn = t if direction == "forward" else 1.0 - t
        b = b_net(x, tn, cond)
        s = s_net(x, tn, cond) * gamma_inv(tn, gamma_type)[:, None, None]
        dW = beta_max * noise[k - 1]
        noise_scale = delta_t * torch.sqrt(2 * epsilon(tn[0], epsilon_type))
        score_eps = score_weight * epsilon(tn[0], epsilon_type)
        new_x = x + (b + score_eps * s) * delta_t if direction == "forward" else x - (b - score_eps * s) * delta_t
        xs.append(new_x + noise_scale * dW)
    return xs[-1], xs


def sde_vs(v_net: Callable, s_net: Callable, x_initial: torch.Tensor, cond: torch.Tensor,
           noise: torch.Tensor, diffuse_step: int = 10, beta_max: float = 0.03,
           gamma_type: str = "2^0.5*t(t-1)", epsilon_type: str = "1-t", score_weight: float = 1.0,
           direction: str = "forward") -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """Velocity-score SDE, Euler–Maruyama (bridge_model.py:334-387); direction='backward' (:356-361, :379-382) evaluates nets and gamma
    schedules at 1 - t and steps x -= (b - w*eps(1-t)*s)*dt — the epsilon inside b stays eps(t), as :369 writes it.

    `v_net(x, t, cond)` / `s_net(x, t, cond)`; `noise[k-1]` is the N(0,1) draw of step k (the
    reference multiplies it by d = beta_max, :372).  Operation order follows the reference so
    the fp32 result matches to rounding."""
    delta_t = float(1.0 / diffuse_step)
    n_steps = int(1.0 / delta_t)
    n = x_initial.shape[0]
    xs = [x_initial]
    for k in range(1, n_steps + 1):
        x = xs[-1]
        t = torch.clip(torch.full((n,), k / n_steps).float(), T_MIN, 1.0 - T_MIN)
        tn = t if direction == "forward" else 1.0 - t
        g, gd = gamma(tn, gamma_type), gamma_der(tn, gamma_type)
        v = v_net(x, tn, cond)
        s = s_net(x, tn, cond)
        gi = gamma_inv(tn, gamma_type)
        s = s * gi[:, None, None]
        gdg = (gd * g)[:, None, None]
        b = v - gdg * s * epsilon(t[0], epsilon_type)
        dW = beta_max * noise[k - 1]
        noise_scale = delta_t * torch.sqrt(2 * epsilon(tn[0], epsilon_type))
        score_eps = score_weight * epsilon(tn[0], epsilon_type)
        new_x = x + (b + score_eps * s) * delta_t if direction == "forward" else x - (b - score_eps * s) * delta_t
        new_x = new_x + noise_scale * dW
        xs.append(new_x)
    return xs[-1], xs
