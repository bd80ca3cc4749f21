"""SigLIP image-token tower: oracle vs the reference wrapper's outputs (G11, CPU) and HIP engine vs G11 (GPU)."""
import numpy as np
import pytest
import torch

from tests import cases
from vlatouch import synth

torch.set_grad_enabled(False)


def G():
    return np.load(f"{cases.GOLDEN}/g11_siglip.npz")


def err(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.mark.parametrize("name", ["tiny", "wide2"])
def test_oracle_matches_reference_tower(name):
    from oracle import siglip
    c = synth.SIGLIP_CONFIGS[name]
    px = cases.siglip_pixels(2, c["image_size"], seed=8)
    out = siglip.siglip_forward(cases.siglip_sd(name), px, heads=c["heads"])
    g = G()
    assert out.shape == g[f"{name}_tokens"].shape
    assert err(out, g[f"{name}_tokens"]) < 5e-5
    assert err(out, g[f"{name}_list"]) < 5e-5                     # the reference's one-image-at-a-time list path agrees


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["tiny", "wide2"])
def test_tower_matches_reference(name, prec):
    from models.multimodal_encoder.siglip_encoder import SiglipVisionTower
    import types
    c = synth.SIGLIP_CONFIGS[name]
    cfg = dict(hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
               image_size=c["image_size"], patch_size=14)
    tower = SiglipVisionTower("synthetic", types.SimpleNamespace(mm_vision_select_feature="patch"), device="cuda:0", precision=prec,
                              state_dict=cases.siglip_sd(name), config=cfg)
    px = cases.siglip_pixels(2, c["image_size"], seed=8).to("cuda:0")
    out = tower(px)
    g = G()
    assert out.shape == g[f"{name}_tokens"].shape and tower.num_patches == out.shape[1] and tower.hidden_size == out.shape[2]
    tol = 2e-4 if prec == "fp32" else 1e-2                        # tokens are O(1..5) after post_layernorm; fp16 storage in low precision
    e = err(out, g[f"{name}_tokens"])
    print(f"[{name} {prec}] max abs err {e:.3e}")
    assert e < tol, e
    lst = tower([px[0], px[1]])
    assert isinstance(lst, list) and err(torch.cat(lst), g[f"{name}_list"]) < tol
    with pytest.raises(NotImplementedError):
        SiglipVisionTower("synthetic", types.SimpleNamespace(mm_vision_select_feature="cls_patch"), device="cuda:0", precision=prec,
                          state_dict=cases.siglip_sd(name), config=cfg)(px)


# ------------------------------------------------------------------ RoboticDiffusionTransformerModel wrapper (franka_model_eef.py)
ARGS = {"common": {"img_history_size": 2, "num_cameras": 3, "state_dim": 128, "action_chunk_size": 8},
        "model": {"lang_token_dim": 96, "img_token_dim": 576, "state_token_dim": 128,
                  "rdt": {"hidden_size": 256, "depth": 2, "num_heads": 4}, "lang_adaptor": "mlp2x_gelu", "img_adaptor": "mlp2x_gelu",
                  "state_adaptor": "mlp3x_gelu",
                  "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": 5, "beta_schedule": "squaredcos_cap_v2",
                                      "prediction_type": "sample", "clip_sample": False}},
        "dataset": {"tokenizer_max_length": 16, "image_aspect_ratio": "pad", "auto_adjust_image_brightness": True}}


def _pil_frames():
    from PIL import Image
    g = np.random.default_rng(5)
    mk = lambda h, w, s: Image.fromarray((g.random((h, w, 3)) * 255 * s).astype(np.uint8))
    return [mk(48, 64, 1.0), None, mk(64, 40, 1.0), mk(56, 56, 0.2), mk(30, 90, 1.0), mk(64, 64, 1.0)]     # wide, missing, tall, dark, ...


def test_wrapper_preprocessing_matches_hf_processor_and_state_packing():
    import types
    from PIL import Image, ImageEnhance
    from transformers import SiglipImageProcessor
    from scripts.franka_model_eef import RoboticDiffusionTransformerModel, SiglipPreprocessor, DEFAULT_STATE_INDICES
    vis = types.SimpleNamespace(config=types.SimpleNamespace(image_size=56), num_patches=16, hidden_size=576, eval=lambda: None)
    pol = types.SimpleNamespace(eval=lambda: None)
    m = RoboticDiffusionTransformerModel(ARGS, device="cpu", dtype=torch.float32, vision_model=vis, policy=pol)
    got = m.preprocess_images(_pil_frames())
    assert got.shape == (6, 3, 56, 56)
    # the same steps with HF's own processor (PIL backend here) on the same padded images
    hf = SiglipImageProcessor(size={"height": 56, "width": 56})
    mean255 = (127, 127, 127)
    exp = []
    for im in _pil_frames():
        if im is None:
            im = Image.fromarray(np.ones((56, 56, 3), dtype=np.uint8) * np.array(mean255, dtype=np.uint8).reshape(1, 1, 3))
        px = np.asarray(im, dtype=np.float64)
        if px.sum() / (px.shape[0] * px.shape[1] * 255.0 * 3) <= 0.15:
            im = ImageEnhance.Brightness(im).enhance(1.75)
        w, h = im.size
        if w != h:
            side = max(w, h)
            sq = Image.new(im.mode, (side, side), mean255)
            sq.paste(im, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
            im = sq
        exp.append(hf.preprocess(im, return_tensors="pt")["pixel_values"][0])
    assert float((got - torch.stack(exp)).abs().max()) < 1e-6
    # 10-d <-> 128-d packing (franka_model_eef.py:167-221)
    j = torch.arange(10, dtype=torch.float32).reshape(1, 1, 10) + 1.0
    st, mask = m._format_joint_to_state(j)
    assert st.shape == (1, 1, 128) and mask.shape == (1, 128) and int(mask.sum()) == 10
    assert torch.allclose(st[0, 0, DEFAULT_STATE_INDICES], j[0, 0] / torch.tensor([1.0] * 9 + [255.0]))
    assert torch.allclose(m._unformat_action_to_joint(st), j)
    with pytest.raises(NotImplementedError):
        m.encode_instruction("wipe the table")


@pytest.mark.gpu
def test_wrapper_step_end_to_end():
    """step() == SigLIP tower on the 6 preprocessed frames -> [1, 6*16, 576] image tokens -> predict_action -> 10 EEF dims."""
    import types
    from models.multimodal_encoder.siglip_encoder import SiglipVisionTower
    from scripts.franka_model_eef import RoboticDiffusionTransformerModel
    c = synth.SIGLIP_CONFIGS["tiny"]
    cfg = dict(hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
               image_size=c["image_size"], patch_size=14)
    tower = SiglipVisionTower("synthetic", None, device="cuda:0", precision="fp32", state_dict=cases.siglip_sd("tiny"), config=cfg)
    m = RoboticDiffusionTransformerModel(ARGS, device="cuda:0", dtype=torch.float32, control_frequency=10, vision_model=tower)
    assert m.policy.model.img_cond_len == 6 * 16
    g = torch.Generator().manual_seed(3)
    proprio = torch.randn(1, 10, generator=g)
    text = torch.randn(1, 12, 96, generator=g)
    torch.manual_seed(11)
    traj = m.step(proprio, _pil_frames(), text)
    assert traj.shape == (1, 8, 10) and traj.dtype == torch.float32 and torch.isfinite(traj).all()
    # manual composition
    px = m.preprocess_images(_pil_frames()).to("cuda:0")
    img_tokens = tower(px).reshape(1, -1, 576)
    st, mask = m._format_joint_to_state(proprio.to("cuda:0").unsqueeze(0))
    torch.manual_seed(11)
    ref = m.policy.predict_action(lang_tokens=text.to("cuda:0"), lang_attn_mask=torch.ones(1, 12, dtype=torch.bool, device="cuda:0"),
                                  img_tokens=img_tokens, state_tokens=st[:, -1:], action_mask=mask.unsqueeze(1),
                                  ctrl_freqs=torch.tensor([10], device="cuda:0"))
    assert torch.equal(traj, m._unformat_action_to_joint(ref).float())


@pytest.mark.gpu
def test_batched_episode_labelling_equals_per_timestep_steps():
    """vlatouch.label.label_episode (frames encoded once, chunks generated in batches) == the reference's loop of
    policy.step(...) over the 2-deep observation window (create_controller_dataset_episode.py:186-205), timestep by timestep."""
    from models.multimodal_encoder.siglip_encoder import SiglipVisionTower
    from scripts.franka_model_eef import RoboticDiffusionTransformerModel
    from vlatouch.label import label_episode
    c = synth.SIGLIP_CONFIGS["tiny"]
    cfg = dict(hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
               image_size=c["image_size"], patch_size=14)
    tower = SiglipVisionTower("synthetic", None, device="cuda:0", precision="fp32", state_dict=cases.siglip_sd("tiny"), config=cfg)
    m = RoboticDiffusionTransformerModel(ARGS, device="cuda:0", dtype=torch.float32, control_frequency=10, vision_model=tower)
    g = np.random.default_rng(21)
    N = 7
    cam1 = (g.random((N, 56, 56, 3)) * 255).astype(np.uint8)
    cam2 = (g.random((N, 56, 56, 3)) * 255).astype(np.uint8)
    qpos = g.standard_normal((N, 10)).astype(np.float32)
    qpos[:, 9] = g.uniform(0, 255, N)
    text = torch.from_numpy(g.standard_normal((1, 12, 96)).astype(np.float32))
    noise = torch.from_numpy(g.standard_normal((N, 8, 128)).astype(np.float32))
    got = label_episode(m, cam1, cam2, qpos, text, batch=3, noise=noise, encode_batch=4)
    assert got.shape == (N, 8, 10) and got.dtype == np.float32
    # the reference's loop, one timestep at a time through step()'s own code path (noise injected through the generator it draws from)
    from PIL import Image
    for t in range(N):
        imgs = [Image.fromarray(cam1[t - 1]) if t > 0 else None, Image.fromarray(cam2[t - 1]) if t > 0 else None, None,
                Image.fromarray(cam1[t]), Image.fromarray(cam2[t]), None]
        px = m.preprocess_images(imgs).to("cuda:0")
        tokens = tower(px).reshape(1, -1, 576)
        st, mask = m._format_joint_to_state(torch.from_numpy(qpos[t:t + 1]).cuda().unsqueeze(0))
        ref = m.policy.predict_action(lang_tokens=text.cuda(), lang_attn_mask=torch.ones(1, 12, dtype=torch.bool, device="cuda:0"),
                                      img_tokens=tokens, state_tokens=st[:, -1:], action_mask=mask.unsqueeze(1),
                                      ctrl_freqs=torch.tensor([10.0], device="cuda:0"), x_init=noise[t:t + 1].cuda())
        ref = m._unformat_action_to_joint(ref).float().cpu().numpy()[0]
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(got[t] - ref).max()) < 2e-4 * scale, t
