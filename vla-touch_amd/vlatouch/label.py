"""Batched offline VLA-action labelling of an episode (the throughput-bound consumer of the RDT path).

The reference labels one timestep at a time (data/create_controller_dataset_episode.py:186-205): for every t it pushes the two
camera frames into a 2-deep observation window (the third camera is always None, the slot before the first frame is an all-None
dummy, :50-97) and calls `policy.step(proprio=qpos[t], images=[ext_{t-1}, right_{t-1}, None, ext_t, right_t, None], text_embeds)`
(:99-125) — 6 SigLIP forwards and one batch-1 RDT chunk per timestep.  None of that depends on the model's outputs, so here
  * every distinct frame goes through the SigLIP tower ONCE (a frame serves the windows of t and t+1; the None slots are one cached
    background image): 2 image encodes per timestep instead of 6, in large batches;
  * the RDT chunks of `batch` timesteps are generated together (323 vs 32 chunks/s at batch 32 vs 1 on an MI355X).
Frames are expected already padded/resized for SigLIP (the `camera{1,2}_resized` arrays the reference's labeller writes); the
reference's cv2 JPEG round trip (:53-57, "to align with training") is data preparation and is not reproduced (no cv2 here).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch


@torch.no_grad()
def encode_frames(model, frames: Sequence, batch: int = 48) -> torch.Tensor:
    """frames: sequence of HxWx3 uint8 arrays / PIL images / None -> SigLIP tokens [n, num_patches, hidden] on the model's device."""
    from PIL import Image
    out = []
    for i in range(0, len(frames), batch):
        pil = [None if f is None else (f if isinstance(f, Image.Image) else Image.fromarray(np.asarray(f))) for f in frames[i:i + batch]]
        px = model.preprocess_images(pil).to(model.device, dtype=model.dtype)
        out.append(model.vision_model(px))
    return torch.cat(out, dim=0)


@torch.no_grad()
def label_episode(model, frames_cam1: Sequence, frames_cam2: Sequence, qpos, lang_embeddings: torch.Tensor, *, batch: int = 32,
                  noise: Optional[torch.Tensor] = None, encode_batch: int = 48) -> np.ndarray:
    """VLA action chunks for every timestep of an episode: [N, horizon, 10] float32 (gripper back in 0..255), what the reference's
    per-timestep loop stores as `vla_action`.

    model: scripts.franka_model_eef.RoboticDiffusionTransformerModel;  frames_cam{1,2}: N frames each (ext / right-wrist camera);
    qpos [N, 10] (utils_eef.converted_ee_pose_with_gripper);  lang_embeddings [1, L, lang_dim] cached instruction embedding;
    noise: optional [N, horizon, action_dim] start noise per timestep (the reference draws it inside predict_action)."""
    N = len(frames_cam1)
    assert len(frames_cam2) == N and len(qpos) == N
    dev, dt = model.device, model.dtype
    tok1 = encode_frames(model, list(frames_cam1), encode_batch)             # [N, P, D]
    tok2 = encode_frames(model, list(frames_cam2), encode_batch)
    bg = encode_frames(model, [None], 1)[0]                                  # the None slots: one background image
    qpos = torch.as_tensor(np.asarray(qpos), dtype=torch.float32, device=dev)
    text = lang_embeddings.to(dev, dtype=dt)
    P, D = bg.shape
    out = []
    for t0 in range(0, N, batch):
        ts = list(range(t0, min(N, t0 + batch)))
        B = len(ts)
        img = torch.empty(B, 6, P, D, dtype=tok1.dtype, device=dev)
        for j, t in enumerate(ts):
            img[j, 0] = tok1[t - 1] if t > 0 else bg                          # window[-2]: previous frame (all None before the first)
            img[j, 1] = tok2[t - 1] if t > 0 else bg
            img[j, 2] = bg
            img[j, 3] = tok1[t]
            img[j, 4] = tok2[t]
            img[j, 5] = bg
        states, mask = model._format_joint_to_state(qpos[ts].unsqueeze(1))     # [B, 1, 128], [B, 128]
        traj = model.policy.predict_action(
            lang_tokens=text.expand(B, -1, -1).contiguous(), lang_attn_mask=torch.ones(B, text.shape[1], dtype=torch.bool, device=dev),
            img_tokens=img.reshape(B, 6 * P, D), state_tokens=states.to(dt), action_mask=mask.unsqueeze(1).to(dt),
            ctrl_freqs=torch.full((B,), float(model.control_frequency), device=dev),
            **({"x_init": noise[ts].to(dev)} if noise is not None else {}))
        out.append(model._unformat_action_to_joint(traj).to(torch.float32).cpu())
    return torch.cat(out, dim=0).numpy()
