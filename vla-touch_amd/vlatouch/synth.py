"""Deterministic synthetic weights and inputs.

There is no network on either box, so no pretrained DINOv2 / RDT / controller
checkpoint exists.  Every tensor is regenerated from its *state-dict key and
shape* by a counter-based generator (numpy Philox keyed by a SHA-256 of the
name), so the golden-vector generator (tools/make_golden.py, which fills the
imported reference modules), the oracle, the HIP path and bench.py all see
bit-identical weights without shipping them.

Scales are chosen so activations stay O(1) through deep stacks and so that
every parameter is non-trivial (norm gains != 1, biases != 0): a kernel that
drops an affine term fails parity.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Iterable, Mapping, Sequence, Tuple

import numpy as np

__all__ = ["swiglu_hidden", "tensor", "fill_state_dict", "inputs_rng", "unet_shapes", "si_net_shapes",
           "dinov2_shapes", "DINOV2_CONFIGS", "state_encoder_shapes", "lstm_controller_shapes",
           "rdt_runner_shapes"]


def _rng(name: str, salt: str = "") -> np.random.Generator:
    h = hashlib.sha256((salt + "|" + name).encode()).digest()
    key = np.frombuffer(h[:16], dtype=np.uint64)
    return np.random.Generator(np.random.Philox(key=key))


def tensor(name: str, shape: Sequence[int], salt: str = "") -> np.ndarray:
    """fp32 tensor for state-dict key `name` with `shape` (rule depends on name+shape only)."""
    shape = tuple(int(s) for s in shape)
    g = _rng(name, salt)
    z = g.standard_normal(size=shape, dtype=np.float32)
    leaf = name.split(".")[-1]
    if any(k in name for k in ("pos_embed", "position_embeddings", "cls_token", "mask_token")):
        return (0.2 * z).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (z / np.sqrt(max(fan_in, 1))).astype(np.float32)
    # 1-D tensors: norm gains / LayerScale vs biases
    if leaf in ("weight", "lambda1"):
        return (1.0 + 0.1 * z).astype(np.float32)
    return (0.1 * z).astype(np.float32)


def fill_state_dict(shapes: Mapping[str, Sequence[int]], prefix: str = "", salt: str = "") -> Dict[str, np.ndarray]:
    return {k: tensor(prefix + k, s, salt) for k, s in shapes.items()}


def fill_state_dict_device(shapes: Mapping[str, Sequence[int]], device, dtype, seed: int = 0):
    """Same scale rules as `tensor`, drawn with the DEVICE generator straight into HBM in `dtype` — for billion-parameter
    random-init models (bench.py's RDT-1B), where a host-side numpy fill + upload would take minutes.  Not bit-identical to
    `tensor` (different generator); used only where no golden vector depends on the values."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        z = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        leaf = name.split(".")[-1]
        if any(k in name for k in ("pos_embed", "position_embeddings", "cls_token", "mask_token")):
            z.mul_(0.2)
        elif len(shape) >= 2:
            z.mul_(1.0 / float(np.sqrt(max(int(np.prod(shape[1:])), 1))))
        elif leaf in ("weight", "lambda1"):
            z.mul_(0.1).add_(1.0)
        else:
            z.mul_(0.1)
        out[name] = z.to(dtype)
    return out


def inputs_rng(seed: int = 1234) -> np.random.Generator:
    """synthetic inputs come from numpy PCG64(seed)."""
    return np.random.Generator(np.random.PCG64(seed))


# --------------------------------------------------------------------------------------
# Shape tables (the checkpoint key maps of SURVEY Appendix A).  These are the build's own
# statement of the boundary; tests/test_shapes_vs_reference.py checks them against the
# key/shape lists captured from the imported reference modules (tests/golden/shapes.json).
# --------------------------------------------------------------------------------------

def _resblock(p: str, cin: int, cout: int, cond: int, k: int) -> Dict[str, Tuple[int, ...]]:
    d = {
        f"{p}.blocks.0.block.0.weight": (cout, cin, k), f"{p}.blocks.0.block.0.bias": (cout,),
        f"{p}.blocks.0.block.1.weight": (cout,), f"{p}.blocks.0.block.1.bias": (cout,),
        f"{p}.blocks.1.block.0.weight": (cout, cout, k), f"{p}.blocks.1.block.0.bias": (cout,),
        f"{p}.blocks.1.block.1.weight": (cout,), f"{p}.blocks.1.block.1.bias": (cout,),
        f"{p}.cond_encoder.1.weight": (2 * cout, cond), f"{p}.cond_encoder.1.bias": (2 * cout,),
    }
    if cin != cout:
        d[f"{p}.residual_conv.weight"] = (cout, cin, 1)
        d[f"{p}.residual_conv.bias"] = (cout,)
    return d


def unet_shapes(input_dim: int = 10, global_cond_dim: int = 256, dsed: int = 256,
                down_dims: Sequence[int] = (256, 512, 512), k: int = 5) -> Dict[str, Tuple[int, ...]]:
    """Key/shape map of one conditional 1-D U-Net (reference conditional_unet_1D.py:108-192);
    registration order mid, step-encoder, up, down, final (conditional_unet_1D.py:143,185-188)."""
    cond = dsed + global_cond_dim
    all_dims = [input_dim] + list(down_dims)
    in_out = list(zip(all_dims[:-1], all_dims[1:]))
    mid = all_dims[-1]
    d: Dict[str, Tuple[int, ...]] = {}
    for i in range(2):
        d.update(_resblock(f"mid_modules.{i}", mid, mid, cond, k))
    d["diffusion_step_encoder.1.weight"] = (dsed * 4, dsed)
    d["diffusion_step_encoder.1.bias"] = (dsed * 4,)
    d["diffusion_step_encoder.3.weight"] = (dsed, dsed * 4)
    d["diffusion_step_encoder.3.bias"] = (dsed,)
    for ind, (din, dout) in enumerate(reversed(in_out[1:])):
        d.update(_resblock(f"up_modules.{ind}.0", dout * 2, din, cond, k))
        d.update(_resblock(f"up_modules.{ind}.1", din, din, cond, k))
        # `is_last` in the reference is never true in the up path (ind >= len(in_out)-1 with
        # ind < len(in_out)-1), so every level has a ConvTranspose1d(k=4) (weight is (in,out,k)).
        d[f"up_modules.{ind}.2.conv.weight"] = (din, din, 4)
        d[f"up_modules.{ind}.2.conv.bias"] = (din,)
    for ind, (din, dout) in enumerate(in_out):
        d.update(_resblock(f"down_modules.{ind}.0", din, dout, cond, k))
        d.update(_resblock(f"down_modules.{ind}.1", dout, dout, cond, k))
        if ind < len(in_out) - 1:
            d[f"down_modules.{ind}.2.conv.weight"] = (dout, dout, 3)
            d[f"down_modules.{ind}.2.conv.bias"] = (dout,)
    s = down_dims[0]
    d["final_conv.0.block.0.weight"] = (s, s, k)
    d["final_conv.0.block.0.bias"] = (s,)
    d["final_conv.0.block.1.weight"] = (s,)
    d["final_conv.0.block.1.bias"] = (s,)
    d["final_conv.1.weight"] = (input_dim, s, 1)
    d["final_conv.1.bias"] = (input_dim,)
    return d


def si_net_shapes(input_dim: int = 10, global_cond_dim: int = 256, **kw) -> Dict[str, Tuple[int, ...]]:
    """b_net, v_net, s_net in registration order (conditional_unet_1D_si.py:25-50)."""
    one = unet_shapes(input_dim, global_cond_dim, **kw)
    d: Dict[str, Tuple[int, ...]] = {}
    for net in ("b_net", "v_net", "s_net"):
        for kname, shp in one.items():
            d[f"{net}.{kname}"] = shp
    return d


DINOV2_CONFIGS = {
    # name fragment -> (hidden, layers, heads)  (visual_encoder.py:31-46; HF configs)
    "small": dict(hidden=384, layers=12, heads=6),
    "base": dict(hidden=768, layers=12, heads=12),
    "large": dict(hidden=1024, layers=24, heads=16),
    "giant": dict(hidden=1536, layers=40, heads=24, swiglu=True),          # SwiGLU FFN (HF Dinov2SwiGLUFFN): weights_in [2 F, D], weights_out [D, F]
    "giant-l4": dict(hidden=1536, layers=4, heads=24, swiglu=True),        # the first 4 blocks of giant (same tensor names): a light fixture of the SwiGLU arithmetic
}


def swiglu_hidden(hidden: int, mlp_ratio: int = 4) -> int:
    """Width F of HF Dinov2SwiGLUFFN: (int(hidden * mlp_ratio * 2 / 3) + 7) // 8 * 8 (1536 -> 4096)."""
    return (int(hidden * mlp_ratio * 2 / 3) + 7) // 8 * 8


def dinov2_shapes(hidden: int, layers: int, heads: int = 0, image_size: int = 518, patch: int = 14, swiglu: bool = False) -> Dict[str, Tuple[int, ...]]:
    """HF Dinov2Model key map."""
    n_pos = (image_size // patch) ** 2 + 1
    D = hidden
    d: Dict[str, Tuple[int, ...]] = {
        "embeddings.cls_token": (1, 1, D),
        "embeddings.mask_token": (1, D),
        "embeddings.position_embeddings": (1, n_pos, D),
        "embeddings.patch_embeddings.projection.weight": (D, 3, patch, patch),
        "embeddings.patch_embeddings.projection.bias": (D,),
    }
    for i in range(layers):
        p = f"encoder.layer.{i}"
        d[f"{p}.norm1.weight"] = (D,)
        d[f"{p}.norm1.bias"] = (D,)
        for n in ("query", "key", "value"):
            d[f"{p}.attention.attention.{n}.weight"] = (D, D)
            d[f"{p}.attention.attention.{n}.bias"] = (D,)
        d[f"{p}.attention.output.dense.weight"] = (D, D)
        d[f"{p}.attention.output.dense.bias"] = (D,)
        d[f"{p}.layer_scale1.lambda1"] = (D,)
        d[f"{p}.norm2.weight"] = (D,)
        d[f"{p}.norm2.bias"] = (D,)
        if swiglu:
            Fh = swiglu_hidden(D)
            d[f"{p}.mlp.weights_in.weight"] = (2 * Fh, D)
            d[f"{p}.mlp.weights_in.bias"] = (2 * Fh,)
            d[f"{p}.mlp.weights_out.weight"] = (D, Fh)
            d[f"{p}.mlp.weights_out.bias"] = (D,)
        else:
            d[f"{p}.mlp.fc1.weight"] = (4 * D, D)
            d[f"{p}.mlp.fc1.bias"] = (4 * D,)
            d[f"{p}.mlp.fc2.weight"] = (D, 4 * D)
            d[f"{p}.mlp.fc2.bias"] = (D,)
        d[f"{p}.layer_scale2.lambda1"] = (D,)
    d["layernorm.weight"] = (D,)
    d["layernorm.bias"] = (D,)
    return d


def _mlp3(din: int, h: int, dout: int) -> Dict[str, Tuple[int, ...]]:
    return {"0.weight": (h, din), "0.bias": (h,), "2.weight": (h, h), "2.bias": (h,),
            "4.weight": (dout, h), "4.bias": (dout,)}


def state_encoder_shapes(obs_dim: int, hidden: int = 256) -> Dict[str, Tuple[int, ...]]:
    """bridge_controller.py:42-48 (Linear-GELU-Linear-GELU-Linear)."""
    return _mlp3(obs_dim, hidden, hidden)


def force_decoder_shapes(hidden: int = 256, force_dim: int = 3) -> Dict[str, Tuple[int, ...]]:
    """bridge_controller.py:50-56 (loaded, unused at inference)."""
    return {"0.weight": (hidden, hidden), "0.bias": (hidden,), "2.weight": (hidden // 2, hidden),
            "2.bias": (hidden // 2,), "4.weight": (force_dim, hidden // 2), "4.bias": (force_dim,)}


def lstm_controller_shapes(latent: int, state_dim: int = 10, hidden: int = 256, layers: int = 2,
                           force_dim: int = 3) -> Dict[str, Dict[str, Tuple[int, ...]]]:
    """tactile_controller.pt['modules'] key map (lstm_step_controller.py:44-82)."""
    obs_dim = 2 * latent + state_dim
    lstm_in = hidden // 2 + state_dim
    lstm: Dict[str, Tuple[int, ...]] = {}
    for l in range(layers):
        i = lstm_in if l == 0 else hidden
        lstm[f"weight_ih_l{l}"] = (4 * hidden, i)
        lstm[f"weight_hh_l{l}"] = (4 * hidden, hidden)
        lstm[f"bias_ih_l{l}"] = (4 * hidden,)
        lstm[f"bias_hh_l{l}"] = (4 * hidden,)
    return {
        "obs_encoder": _mlp3(obs_dim, hidden, hidden),
        "force_encoder": {"0.weight": (hidden // 2, force_dim), "0.bias": (hidden // 2,),
                          "2.weight": (hidden // 2, hidden // 2), "2.bias": (hidden // 2,)},
        "lstm": lstm,
        "output_head": {"0.weight": (hidden, 2 * hidden), "0.bias": (hidden,),
                        "1.weight": (hidden,), "1.bias": (hidden,),
                        "4.weight": (state_dim, hidden), "4.bias": (state_dim,)},
    }


def _adaptor_shapes(prefix: str, kind: str, din: int, dout: int) -> Dict[str, Tuple[int, ...]]:
    import re
    if kind == "linear":
        return {f"{prefix}.weight": (dout, din), f"{prefix}.bias": (dout,)}
    m = re.match(r"^mlp(\d+)x_gelu$", kind)
    if not m:
        raise ValueError(f"Unknown projector type: {kind}")
    depth = int(m.group(1))
    d = {f"{prefix}.0.weight": (dout, din), f"{prefix}.0.bias": (dout,)}
    for i in range(1, depth):
        d[f"{prefix}.{2 * i}.weight"] = (dout, dout)
        d[f"{prefix}.{2 * i}.bias"] = (dout,)
    return d


def rdt_runner_shapes(*, hidden: int, depth: int, heads: int, horizon: int, action_dim: int,
                      lang_token_dim: int, img_token_dim: int, state_token_dim: int,
                      max_lang_cond_len: int, img_cond_len: int,
                      lang_adaptor: str = "mlp2x_gelu", img_adaptor: str = "mlp2x_gelu",
                      state_adaptor: str = "mlp3x_gelu") -> Dict[str, Tuple[int, ...]]:
    """RDTRunner state-dict key map (models/rdt/model.py:48-64, blocks.py:33-37,
    91-97,149-165,192-197; rdt_runner.py:41-60)."""
    D, hd = hidden, hidden // heads
    d: Dict[str, Tuple[int, ...]] = {
        "model.x_pos_embed": (1, horizon + 3, D),
        "model.lang_cond_pos_embed": (1, max_lang_cond_len, D),
        "model.img_cond_pos_embed": (1, img_cond_len, D),
    }
    for e in ("t_embedder", "freq_embedder"):
        d[f"model.{e}.mlp.0.weight"] = (D, 256)
        d[f"model.{e}.mlp.0.bias"] = (D,)
        d[f"model.{e}.mlp.2.weight"] = (D, D)
        d[f"model.{e}.mlp.2.bias"] = (D,)
    for i in range(depth):
        p = f"model.blocks.{i}"
        d[f"{p}.norm1.weight"] = (D,)
        d[f"{p}.attn.qkv.weight"] = (3 * D, D)
        d[f"{p}.attn.qkv.bias"] = (3 * D,)
        d[f"{p}.attn.q_norm.weight"] = (hd,)
        d[f"{p}.attn.k_norm.weight"] = (hd,)
        d[f"{p}.attn.proj.weight"] = (D, D)
        d[f"{p}.attn.proj.bias"] = (D,)
        d[f"{p}.cross_attn.q.weight"] = (D, D)
        d[f"{p}.cross_attn.q.bias"] = (D,)
        d[f"{p}.cross_attn.kv.weight"] = (2 * D, D)
        d[f"{p}.cross_attn.kv.bias"] = (2 * D,)
        d[f"{p}.cross_attn.q_norm.weight"] = (hd,)
        d[f"{p}.cross_attn.k_norm.weight"] = (hd,)
        d[f"{p}.cross_attn.proj.weight"] = (D, D)
        d[f"{p}.cross_attn.proj.bias"] = (D,)
        d[f"{p}.norm2.weight"] = (D,)
        d[f"{p}.ffn.fc1.weight"] = (D, D)
        d[f"{p}.ffn.fc1.bias"] = (D,)
        d[f"{p}.ffn.fc2.weight"] = (D, D)
        d[f"{p}.ffn.fc2.bias"] = (D,)
        d[f"{p}.norm3.weight"] = (D,)
    d["model.final_layer.norm_final.weight"] = (D,)
    d["model.final_layer.ffn_final.fc1.weight"] = (D, D)
    d["model.final_layer.ffn_final.fc1.bias"] = (D,)
    d["model.final_layer.ffn_final.fc2.weight"] = (action_dim, D)
    d["model.final_layer.ffn_final.fc2.bias"] = (action_dim,)
    d.update(_adaptor_shapes("lang_adaptor", lang_adaptor, lang_token_dim, D))
    d.update(_adaptor_shapes("img_adaptor", img_adaptor, img_token_dim, D))
    d.update(_adaptor_shapes("state_adaptor", state_adaptor, state_token_dim * 2, D))
    return d


SIGLIP_CONFIGS = {
    # google/siglip-so400m-patch14-384 (the RDT image tower, models/multimodal_encoder/siglip_encoder.py): 16 heads of 72
    "so400m": dict(hidden=1152, layers=27, heads=16, inter=4304, image_size=384),
    # test-size variants that keep the 72-wide heads and a hidden size the MFMA kernels accept (multiple of 64)
    "tiny": dict(hidden=576, layers=2, heads=8, inter=1072, image_size=56),
    "wide2": dict(hidden=1152, layers=2, heads=16, inter=4304, image_size=98),
}


def siglip_shapes(hidden: int, layers: int, inter: int, image_size: int = 384, patch: int = 14, **_) -> Dict[str, Tuple[int, ...]]:
    """HF SiglipVisionModel key map (transformers 5.x spelling, no `vision_model.` prefix), without the pooling head: the RDT
    tower returns last_hidden_state (siglip_encoder.py:33-35)."""
    D = hidden
    n_pos = (image_size // patch) ** 2
    d: Dict[str, Tuple[int, ...]] = {
        "embeddings.patch_embedding.weight": (D, 3, patch, patch),
        "embeddings.patch_embedding.bias": (D,),
        "embeddings.position_embedding.weight": (n_pos, D),
    }
    for i in range(layers):
        p = f"encoder.layers.{i}"
        d[f"{p}.layer_norm1.weight"] = (D,)
        d[f"{p}.layer_norm1.bias"] = (D,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            d[f"{p}.self_attn.{n}.weight"] = (D, D)
            d[f"{p}.self_attn.{n}.bias"] = (D,)
        d[f"{p}.layer_norm2.weight"] = (D,)
        d[f"{p}.layer_norm2.bias"] = (D,)
        d[f"{p}.mlp.fc1.weight"] = (inter, D)
        d[f"{p}.mlp.fc1.bias"] = (inter,)
        d[f"{p}.mlp.fc2.weight"] = (D, inter)
        d[f"{p}.mlp.fc2.bias"] = (D,)
    d["post_layernorm.weight"] = (D,)
    d["post_layernorm.bias"] = (D,)
    return d


# ------------------------------------------------------------------ a synthetic controller (bench / smoke / tests)
MODEL_ARGS = {
    'interpolant_type': 'linear', 'gamma_type': '2^0.5*t(t-1)', 'epsilon_type': '1-t', 'prior_policy': 'vla',
    'beta_max': 0.03, 'sde_type': 'vs', 'action_dim': 10, 'obs_dim': 256, 'obs_horizon': 1,
    'net_type': 'unet1D_si', 'pretrain': False, 'context_frames': 2, 'horizon': 16,
}


def torch_state_dict(shapes, prefix: str = "", salt: str = ""):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in fill_state_dict(shapes, prefix, salt).items()}


def unit_stats():
    import torch
    z, o = torch.zeros(10), torch.ones(10)
    return dict(action_mins=z.clone(), action_maxs=o.clone(), vla_mins=z.clone(), vla_maxs=o.clone(), action_range=o.clone(), vla_range=o.clone())


def build_controller(cls, precision: str, device="cuda:0", size: str = "small", stats=None, force_dim: int = 3, **kw):
    """A DiffusionController (the mirror class `cls`) filled with the deterministic synthetic weights the goldens were made
    with (no checkpoints exist offline): raw `net` params = salt "", EMA shadow params = salt "ema" — a sampler that
    forgets to run under the EMA weights fails the goldens.  `force_dim` = width of the tactile vector m_t (reference default 3 =
    the marker tracker's force estimate, bridge_controller.py:25; BASELINE's synthetic workload names a 64-d tactile vector)."""
    c = DINOV2_CONFIGS[size]
    ctrl = cls(state_dim=10, hidden_dim=256, image_model_path=f"facebook/dinov2-{size}", diffusion_steps=10, device=device,
               model_args=dict(MODEL_ARGS), use_force=True, force_dim=force_dim, precision=precision,
               image_state_dict=torch_state_dict(dinov2_shapes(c["hidden"], c["layers"]), prefix=f"dinov2-{size}."), **kw)
    latent = c["hidden"]
    ctrl.state_encoder.load_state_dict(torch_state_dict(state_encoder_shapes(2 * latent + 10 + force_dim), prefix="state_encoder."))
    ctrl.force_decoder.load_state_dict(torch_state_dict(force_decoder_shapes(force_dim=force_dim), prefix="force_decoder."))
    ctrl.diffusion_model.net.load_state_dict(torch_state_dict(si_net_shapes(10, 256), prefix="si.", salt=""))
    ema = torch_state_dict(si_net_shapes(10, 256), prefix="si.", salt="ema")
    ctrl.diffusion_model.ema.load_state_dict({"decay": 0.75, "num_updates": 0, "collected_params": None,
                                              "shadow_params": [ema[k] for k in ctrl.diffusion_model.net.state_dict().keys()]})
    ctrl.diffusion_model.net.to(device)
    ctrl.diffusion_model.ema.to(device)
    ctrl.state_encoder.to(device)
    ctrl.stats = {k: v.to(device) for k, v in (stats if stats is not None else unit_stats()).items()}
    return ctrl


# ------------------------------------------------------------------ synthetic GelSight frames (marker tracker inputs)
def synth_gel_frame(rng, shift=(0.0, 0.0), bulge: float = 0.0, H: int = 240, W: int = 320, rows: int = 7, cols: int = 9) -> np.ndarray:
    """A GelSight-like BGR frame: bright gel with an illumination gradient and sensor noise, a rows x cols grid of dark
    round markers displaced by a rigid shift plus a radial bulge (contact)."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    base = 150 + 40 * (xx / W) - 30 * (yy / H)
    img = np.stack([base * 0.9, base, base * 1.05], axis=-1)
    gx = np.linspace(28, W - 28, cols)
    gy = np.linspace(24, H - 24, rows)
    cx0, cy0 = W / 2, H / 2
    k = 0
    for y in gy:
        for x in gx:
            r2 = ((x - cx0) ** 2 + (y - cy0) ** 2) / (cx0 ** 2 + cy0 ** 2)
            px = x + shift[0] + bulge * (x - cx0) / cx0 * np.exp(-3 * r2)
            py = y + shift[1] + bulge * (y - cy0) / cy0 * np.exp(-3 * r2)
            d2 = (xx - px) ** 2 + 1.0 * (yy - py) ** 2 * (1.0 + 0.15 * ((k * 7) % 5 - 2) / 2)      # slightly elliptic, per-marker size
            sig = 2.6 + 0.25 * ((k * 5) % 6)
            img *= (1 - 0.75 * np.exp(-d2 / (2 * sig ** 2)))[..., None]
            k += 1
    img += rng.normal(0, 3.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
