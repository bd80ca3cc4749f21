"""Stochastic-interpolant sampler — mirror of the reference's
VLA/residual_controller/bridge/bridge_model.py:28-446 (inference surface).

Kept from the reference: the constructor / `load_model_args` contract, the string-keyed schedules
(`epsilon`, `gamma`, `gamma_der`, `gamma_inv`, NotImplementedError on unknown keys, :59-101), `sample(x_prior,
cond, diffuse_step=10, recod_traj=False)` (:259-279) which samples under the EMA shadow weights, the
`sde_vs` / `sde_bs` integrators (:281-387, forward direction), `load_model` / `save_model` and the
`bridge_model.pt` = {"net", "ema"} checkpoint format (:421-446), `train()` / `eval()`.
Not kept (training-only, out of scope per SURVEY §2 row 2): the losses, q_sample, interpolant().

The whole n-step Euler–Maruyama loop is ONE call into the HIP engine (vt_si_sample): v_net and s_net are
evaluated in grouped launches, the SDE update is a fused element-wise kernel.  The reference draws its
Gaussian noise with torch.randn_like inside the loop; here the same draws are made up-front
(`noise=` lets a caller inject them — that is how parity with the reference is tested).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from vlatouch.module import ExponentialMovingAverage, default_precision
from residual_controller.bridge.networks.conditional_unet_1D_si import InterpolantsConditionalUnet1D

_GAMMA = {"2^0.5*t(t-1)": 0, "(2t(t-1))^0.5": 1, "(1-t)^2(2t)^0.5": 2}
_EPS = {"1-t": 0, "t(t-1)": 1, "1-sqrt(t)": 2, "1-t^2": 3, "0": 4}

_R2 = 1.4142       # the reference's literal for sqrt(2) (bridge_model.py:75-100), kept to the digit: goldens compare schedules at 1e-6
_TINY = 1e-4       # its guard against 1/0 at the interval ends
# epsilon_type -> eps(t)
_EPSILON_OF_T = {
    "1-t": lambda t, si: 1.0 - t,
    "t(t-1)": lambda t, si: t * (1 - t),
    "1-sqrt(t)": lambda t, si: 1.0 - t.sqrt(),
    "1-t^2": lambda t, si: 1.0 - t * t,
    "0": lambda t, si: t * 0.0,
}
# gamma_type -> (gamma(t), d gamma / dt, the guarded denominator whose reciprocal — clamped to [0, gamma_inv_max] — is gamma_inv(t))
_GAMMA_OF_T = {
    "2^0.5*t(t-1)": lambda t, si: (_R2 * t * (1 - t), _R2 * (1 - 2 * t), _R2 * t * (1 - t) + _TINY),
    "(2t(t-1))^0.5": lambda t, si: (_R2 * (t * (1 - t)).sqrt(), (1 - 2 * t) / (2 * (t - t * t) + _TINY).sqrt(), _R2 * (t * (1 - t) + _TINY).sqrt()),
    "(1-t)^2(2t)^0.5": lambda t, si: (_R2 * (1 - t) ** 2 * t.sqrt(),
                                      _R2 * (2 * (t - 1) * t.sqrt() + (1 - t) ** 2 / (2.0 * (t + _TINY).sqrt())),
                                      _R2 * (1 - t) ** 2 * t.sqrt() + _TINY),
}


class StochasticInterpolants:
    def __init__(self, model_args=None, precision: Optional[str] = None):
        self.precision = precision or default_precision()
        self.net = None
        self.ema = None
        self._sampler = None
        if model_args:
            self.load_model_args(model_args)
        else:
            print("init SI without model args")

    def load_model_args(self, model_args):
        self.interpolant_type = model_args['interpolant_type']
        self.gamma_type = model_args['gamma_type']
        self.epsilon_type = model_args['epsilon_type']
        self.prior_policy = model_args['prior_policy']
        self.d = model_args['beta_max']
        self.t_min = 0.001
        self.gamma_inv_max = 200.0
        self.net = None
        self.ema = None
        self.prior_model = None
        self._sampler = None
        self.sde_type = model_args['sde_type'] if 'sde_type' in model_args else 'vs'

    # ---- schedules: thin string-keyed lookups into the tables below (the reference's four if/elif methods, bridge_model.py:59-101, as data).
    #      Dead on the sampling path — csrc/vt_unet.hip (si_schedule) evaluates the same formulas per SDE step from the codes in _GAMMA / _EPS;
    #      kept because they are public methods of the reference class.  Unknown key -> NotImplementedError, as there.
    def _schedule(self, table, key, t):
        try:
            f = table[key]
        except KeyError:
            raise NotImplementedError(key) from None
        return f(t, self)      # `t` as the caller gave it: a Python float stays a float where the formula has no torch call, exactly as in the reference

    def epsilon(self, t):
        return self._schedule(_EPSILON_OF_T, self.epsilon_type, t)

    def gamma(self, t):
        return self._schedule(_GAMMA_OF_T, self.gamma_type, t)[0]

    def gamma_der(self, t):
        return self._schedule(_GAMMA_OF_T, self.gamma_type, t)[1]

    def gamma_inv(self, t):
        g_floor = self._schedule(_GAMMA_OF_T, self.gamma_type, t)[2]
        return (1.0 / g_floor).clamp(0.0, self.gamma_inv_max)

    # ---- sampling
    def _sampler_engine(self, nets, device):
        """U-Net pair engine built from the EMA shadow parameters (what `with self.ema.average_parameters()` makes the
        reference sample with, bridge_model.py:267).  Cached until the EMA state or the net is reloaded."""
        key = (tuple(nets), self.ema.version, self.net.version, str(device))
        if self._sampler is None or self._sampler[0] != key:
            from vlatouch.engine import UNetEngine
            names = list(self.net._params.keys())
            shadow = dict(zip(names, self.ema.shadow_params))
            sds = [{k[len(n) + 1:]: v for k, v in shadow.items() if k.startswith(n + ".")} for n in nets]
            eng = UNetEngine(sds, device=device, precision=self.precision, **self.net.cfg)
            self._sampler = (key, eng)
        return self._sampler[1]

    def _run(self, nets, sde_code, x_initial, cond, delta_t, score_weight, direction, noise, record=True, own_x=False):
        if direction not in ('forward', 'backward'):
            raise NotImplementedError
        if self.gamma_type not in _GAMMA or self.epsilon_type not in _EPS:
            raise NotImplementedError
        n_steps = int(1.0 / delta_t)
        dev = x_initial.device if x_initial.device.type == "cuda" else torch.device("cuda")
        if noise is None:   # the reference's `self.d * torch.randn_like(current_x)` draws, made up-front (torch's generator: `torch.manual_seed`
            # governs them as in the reference; a caller that wants no torch kernel in its step passes `noise=` from vlatouch.ops.DeviceRng)
            noise = torch.randn((n_steps,) + tuple(x_initial.shape), dtype=torch.float32, device=dev)
        eng = self._sampler_engine(nets, dev)
        res = eng.sample(x_initial, cond, noise, n_steps, float(self.d), record=record,
                         gamma_type=_GAMMA[self.gamma_type], epsilon_type=_EPS[self.epsilon_type], sde_type=sde_code,
                         backward=direction == 'backward', score_weight=float(score_weight), own_x0=own_x)
        if not record:       # the hot path: no trajectory buffer, no per-step copies
            return res, None
        xT, traj = res
        return xT, [traj[i] for i in range(traj.shape[0])]

    def sde_vs(self, v_net=None, s_net=None, x_initial=None, cond=None, delta_t=0.025, score_weight=1.0, direction='forward', noise=None, record=True, _own_x=False):
        return self._run(("v_net", "s_net"), 0, x_initial, cond, delta_t, score_weight, direction, noise, record, _own_x)

    def sde_bs(self, b_net=None, s_net=None, x_initial=None, cond=None, delta_t=0.025, score_weight=1.0, direction='forward', noise=None, record=True, _own_x=False):
        return self._run(("b_net", "s_net"), 1, x_initial, cond, delta_t, score_weight, direction, noise, record, _own_x)

    def sample(self, x_prior, cond, diffuse_step=10, recod_traj=False, noise=None, _own_prior=False):
        """x_prior (batch, T, dim) normalised prior actions, cond (batch, obs_dim) -> refined normalised actions.
        `_own_prior=True` (extension, used by DiffusionController.predict whose x_prior is a temporary): the integrator may run in place
        on x_prior instead of on a copy — one runtime copy kernel less per step."""
        with torch.no_grad():
            if self.sde_type == 'vs':
                x_target, x_target_traj = self.sde_vs(x_initial=x_prior, cond=cond, delta_t=float(1.0 / diffuse_step), noise=noise, record=recod_traj, _own_x=_own_prior)
            elif self.sde_type == 'bs':
                x_target, x_target_traj = self.sde_bs(x_initial=x_prior, cond=cond, delta_t=float(1.0 / diffuse_step), noise=noise, record=recod_traj, _own_x=_own_prior)
            else:
                raise NotImplementedError
        if recod_traj:
            return x_target, x_target_traj
        return x_target

    def get_loss(self, batch_dict, device, t=None, z=None, backward=True):
        """The three interpolant losses on one batch (bridge_model.py:221-247): batch_dict['obs_cond' | 'vla_act' | 'expert_act'] ->
        (loss, {'v_loss','s_loss','b_loss'}) as 0-d tensors.  The reference returns an autograd graph; here the backward pass runs
        inside the call (`backward=True`) and leaves the gradients in `self.loss_trainer` (a `vlatouch.train.SITrainer` over the
        current net parameters), whose `optimizer_step()` applies AdamW + EMA.  `t` [B] / `z` [B,T,D] replace the reference's
        in-function `torch.rand` / `torch.randn_like` draws when given.  Training the observation MLP as well goes through
        `residual_controller.bridge_train.DiffusionControllerTrainer`."""
        from vlatouch.train import SITrainer
        dev = torch.device(device)
        key = (self.net.version, str(dev))
        if getattr(self, "_loss_trainer_key", None) != key:
            self.loss_trainer = SITrainer(self.net.state_dict(), None, gamma_type=self.gamma_type, interpolant_type=self.interpolant_type,
                                          beta_max=self.d, ema_decay=self.ema.decay, device=dev)
            self._loss_trainer_key = key
        x0 = torch.as_tensor(batch_dict['vla_act'])
        if t is None:
            t = torch.rand(x0.shape[0], device=dev)
        if z is None:
            z = torch.randn(tuple(x0.shape), device=dev)
        loss, info = self.loss_trainer.get_loss(batch_dict['obs_cond'], x0, batch_dict['expert_act'], t, z, backward=backward)
        as_t = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
        return as_t(loss), {k: as_t(v) for k, v in info.items()}

    def train(self):
        if self.net is not None:
            self.net.train()
        return self

    def eval(self):
        if self.net is not None:
            self.net.eval()
        return self

    def load_model(self, model_args, device):
        self.load_model_args(model_args)
        if model_args['net_type'] == 'unet1D_si':
            self.net = InterpolantsConditionalUnet1D(
                input_dim=model_args['action_dim'],
                global_cond_dim=model_args['obs_dim'] * model_args['obs_horizon'],
                precision=self.precision,
            )
        else:
            raise NotImplementedError
        self.ema = ExponentialMovingAverage(self.net.parameters(), decay=0.75).bind(self.net)
        if model_args['pretrain']:
            checkpoint = torch.load(os.path.join(model_args['ckpt_path'], "bridge_model.pt"), map_location="cpu", weights_only=False)
            self.net.load_state_dict(checkpoint['net'])
            self.ema.load_state_dict(checkpoint["ema"])
        self.net.to(device)
        self.ema.to(device)
        self._sampler = None

    def save_model(self, ckpt_path):
        torch.save({"net": {k: v.cpu() for k, v in self.net.state_dict().items()},
                    "ema": {**self.ema.state_dict(), "shadow_params": [p.cpu() for p in self.ema.shadow_params]}},
                   os.path.join(ckpt_path, "bridge_model.pt"))
