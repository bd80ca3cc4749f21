"""Import the read-only reference (/root/reference) in THIS container for golden-vector capture.

Only tools/make_golden.py and the `needs_reference` tests use this; nothing here travels to the
GPU box except as the generated fixtures under tests/golden/.  Absent third-party packages are
replaced by minimal stand-in MODULES (not reference code): torch_ema, diffusers (import-only),
torchvision, h5py, cv2 and a timm shim restating timm.models.vision_transformer.{Attention, Mlp,
RmsNorm, use_fused_attn} (upstream pin timm==1.0.3; un-pinned by the reference => the RDT
golden is pinned to the reference's own files + this shim, see oracle/__init__.py).
"""
from __future__ import annotations

import contextlib
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("VLATOUCH_REFERENCE", "/root/reference")
RMS_MODE = os.environ.get("VLATOUCH_TIMM_RMSNORM", "meansq")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "VLA", "residual_controller"))


class _EMA:
    """torch_ema.ExponentialMovingAverage semantics used by bridge_model.py:267,433-447."""

    def __init__(self, parameters, decay):
        self._params = list(parameters)
        self.decay = decay
        self.num_updates = 0
        self.shadow_params = [p.clone().detach() for p in self._params]
        self.collected_params = None

    def to(self, device=None, dtype=None):
        self.shadow_params = [p.to(device=device) for p in self.shadow_params]
        return self

    def update(self, parameters=None):
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for s, p in zip(self.shadow_params, self._params):
                s.sub_((1.0 - d) * (s - p))

    @contextlib.contextmanager
    def average_parameters(self, parameters=None):
        saved = [p.clone() for p in self._params]
        with torch.no_grad():
            for p, s in zip(self._params, self.shadow_params):
                p.copy_(s)
        try:
            yield
        finally:
            with torch.no_grad():
                for p, s in zip(self._params, saved):
                    p.copy_(s)

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates,
                "shadow_params": self.shadow_params, "collected_params": self.collected_params}

    def load_state_dict(self, sd):
        self.decay = sd["decay"]
        self.num_updates = sd["num_updates"]
        self.shadow_params = [p.clone() for p in sd["shadow_params"]]


class _RmsNorm(nn.Module):
    def __init__(self, channels, eps=1e-6, affine=True, **kw):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels))

    def forward(self, x):
        if RMS_MODE == "var":
            v = torch.var(x, dim=-1, keepdim=True)
        else:
            v = x.pow(2).mean(dim=-1, keepdim=True)
        return x * torch.rsqrt(v + self.eps) * self.weight


class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, **kw):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, attn_drop=0.0, proj_drop=0.0,
                 norm_layer=nn.LayerNorm, **kw):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q, k = self.q_norm(q), self.k_norm(k)
        x = F.scaled_dot_product_attention(q, k, v)
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_done = False


def setup():
    """Install stand-in modules and put the reference on sys.path (idempotent)."""
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError(f"reference not found at {REF}")
    import transformers  # noqa: F401  (must be imported before the torchvision stand-in exists)
    from transformers import Dinov2Model  # noqa: F401
    _stub("torch_ema", ExponentialMovingAverage=_EMA)
    _stub("diffusers")
    _stub("diffusers.schedulers")
    _stub("diffusers.schedulers.scheduling_ddpm", DDPMScheduler=object)
    _stub("diffusers.schedulers.scheduling_dpmsolver_multistep", DPMSolverMultistepScheduler=object)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    _stub("h5py")
    _stub("cv2")
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.vision_transformer", Attention=_Attention, Mlp=_Mlp, RmsNorm=_RmsNorm,
          use_fused_attn=lambda: True)
    # The product's mirror packages have the reference's module names (residual_controller, models): while the
    # reference is imported, take the product directory off sys.path (vlatouch.synth stays cached in sys.modules).
    try:
        import vlatouch.synth  # noqa: F401
    except ImportError:
        pass
    sys.path[:] = [p for p in sys.path if not p.rstrip("/").endswith("vla-touch_amd")]
    for m in list(sys.modules):
        if m.split(".")[0] in ("residual_controller", "models", "bridge"):
            del sys.modules[m]
    vla = os.path.join(REF, "VLA")
    for p in (os.path.join(vla, "residual_controller"), vla):
        if p not in sys.path:
            sys.path.insert(0, p)
    _done = True


def patch_dinov2(build_fn):
    """Make `Dinov2Model.from_pretrained(name)` return build_fn(name) (no hub, no weights on disk)."""
    from transformers import Dinov2Model
    Dinov2Model.from_pretrained = classmethod(lambda cls, name, *a, **k: build_fn(name))


def no_cuda():
    """The reference hard-codes `.cuda()` (bridge_controller.py:241); make it a no-op on this CPU box."""
    torch.Tensor.cuda = lambda self, *a, **k: self
