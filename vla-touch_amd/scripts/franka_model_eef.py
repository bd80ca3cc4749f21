"""Mirror of /root/reference/VLA/scripts/franka_model_eef.py (RoboticDiffusionTransformerModel, :38-313): the RDT wrapper used on
the robot and by the offline labeller — image padding / preprocessing, SigLIP encoding of the 6 frames, 10-d <-> 128-d state
packing, `predict_action`.  Same constructor, `get_policy`, `reset`, `load_pretrained_weights`,
`_format_joint_to_state`, `_unformat_action_to_joint` and `step(proprio, images, text_embeds)`; the SigLIP tower and the RDT
policy are the HIP-backed mirrors.  Differences, all forced by what is absent here:
  * `configs/state_vec.py` is not in the reference tree: the 10 unified-vector slots come from `state_indices=` (default: upstream
    RDT's mapping eef_pos_{x,y,z} = 30..32, eef_angle_0..5 = 33..38, right_gripper_open = 10 [assumed-upstream]);
  * no torchvision: `transforms.Resize` / `ColorJitter(brightness=1.75)` are done with PIL (same resampling / the same
    ImageEnhance.Brightness torchvision itself uses for PIL images); the SiglipImageProcessor arithmetic (bicubic resize to
    image_size, x/255, (x-0.5)/0.5) is restated in `SiglipPreprocessor` and pinned against HF's PIL processor in tests;
  * the T5 text encoder is not loaded (the reference comments it out as well): `encode_instruction` raises."""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch

from models.multimodal_encoder.siglip_encoder import SiglipVisionTower
from models.rdt_runner import RDTRunner

# eef_pos_x, eef_pos_y, eef_pos_z, eef_angle_0..5, right_gripper_open in upstream RDT's configs/state_vec.py [assumed-upstream]
DEFAULT_STATE_INDICES = [30, 31, 32, 33, 34, 35, 36, 37, 38, 10]


class SiglipPreprocessor:
    """SiglipImageProcessor.preprocess for one PIL image: RGB, bicubic resize to (size, size), /255, (x - mean) / std."""

    def __init__(self, size: int = 384, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5)):
        self.size = {"height": size, "width": size}
        self.image_mean, self.image_std = list(image_mean), list(image_std)
        self.rescale_factor = 1 / 255.0

    def preprocess(self, image, return_tensors="pt"):
        from PIL import Image
        image = image.convert("RGB").resize((self.size["width"], self.size["height"]), resample=Image.BICUBIC)
        x = np.asarray(image, dtype=np.float32) * np.float32(self.rescale_factor)
        x = (x - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
        return {"pixel_values": torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None]}


def create_model(args, **kwargs):
    model = RoboticDiffusionTransformerModel(args, **kwargs)
    pretrained = kwargs.get("pretrained", None)
    if pretrained is not None and os.path.isfile(pretrained):
        model.load_pretrained_weights(pretrained)
    return model


class RoboticDiffusionTransformerModel(object):
    def __init__(self, args, device="cuda", dtype=torch.bfloat16, image_size=None, control_frequency=25, pretrained=None,
                 pretrained_vision_encoder_name_or_path=None, *, vision_model=None, policy=None,
                 state_indices: Optional[Sequence[int]] = None):
        self.args = args
        self.dtype = dtype
        self.image_size = image_size
        self.device = device
        self.control_frequency = control_frequency
        self.state_indices = list(state_indices) if state_indices is not None else list(DEFAULT_STATE_INDICES)
        if vision_model is not None:
            self.vision_model = vision_model
        else:
            self.vision_model = SiglipVisionTower(vision_tower=pretrained_vision_encoder_name_or_path, args=None, device=device,
                                                  precision="bf16" if dtype == torch.bfloat16 else "fp32")
        self.image_processor = SiglipPreprocessor(self.vision_model.config.image_size)
        self.policy = policy if policy is not None else self.get_policy(pretrained)
        self.reset()

    def get_policy(self, pretrained):
        if pretrained is None or os.path.isfile(pretrained):
            a = self.args
            n_patch = self.vision_model.num_patches
            img_cond_len = a["common"]["img_history_size"] * a["common"]["num_cameras"] * n_patch
            return RDTRunner(
                action_dim=a["common"]["state_dim"], pred_horizon=a["common"]["action_chunk_size"], config=a["model"],
                lang_token_dim=a["model"]["lang_token_dim"], img_token_dim=a["model"]["img_token_dim"],
                state_token_dim=a["model"]["state_token_dim"], max_lang_cond_len=a["dataset"]["tokenizer_max_length"],
                img_cond_len=img_cond_len,
                img_pos_embed_config=[("image", (a["common"]["img_history_size"], a["common"]["num_cameras"], -n_patch))],
                lang_pos_embed_config=[("lang", -a["dataset"]["tokenizer_max_length"])],
                dtype=self.dtype, device=self.device)
        return RDTRunner.from_pretrained(pretrained)

    def reset(self):
        self.policy.eval()
        self.vision_model.eval()

    def load_pretrained_weights(self, pretrained=None):
        if pretrained is None:
            return
        filename = os.path.basename(pretrained)
        if filename.endswith(".pt"):
            self.policy.load_state_dict(torch.load(pretrained, map_location="cpu")["module"])
        elif filename.endswith(".safetensors"):
            from safetensors.torch import load_file
            self.policy.load_state_dict(load_file(pretrained))
        else:
            raise NotImplementedError(f"Unknown checkpoint format: {pretrained}")

    def encode_instruction(self, instruction, device="cuda"):
        raise NotImplementedError("the T5-XXL text encoder is not part of this build: pass cached instruction embeddings to step()")

    def _format_joint_to_state(self, joints):
        """[B, N, 10] EEF proprioception (gripper 0..255) -> unified state [B, N, state_token_dim] and its element mask [B, dim]."""
        joints = joints / torch.tensor([[[1, 1, 1, 1, 1, 1, 1, 1, 1, 255]]], device=joints.device, dtype=joints.dtype)
        B, N, _ = joints.shape
        dim = self.args["model"]["state_token_dim"]
        state = torch.zeros((B, N, dim), device=joints.device, dtype=joints.dtype)
        state[:, :, self.state_indices] = joints
        mask = torch.zeros((B, dim), device=joints.device, dtype=joints.dtype)
        mask[:, self.state_indices] = 1
        return state, mask

    def _unformat_action_to_joint(self, action):
        joints = action[:, :, self.state_indices]
        return joints * torch.tensor([[[1, 1, 1, 1, 1, 1, 1, 1, 1, 255]]], device=joints.device, dtype=joints.dtype)

    def preprocess_images(self, images):
        """franka_model_eef.py:242-281: None -> background image, optional resize, optional brightness lift, pad to square with
        the processor's mean colour, SiglipImageProcessor.preprocess.  Returns pixel_values [n, 3, S, S] fp32."""
        from PIL import Image, ImageEnhance
        mean255 = tuple(int(x * 255) for x in self.image_processor.image_mean)
        S = self.image_processor.size
        background = Image.fromarray(np.ones((S["height"], S["width"], 3), dtype=np.uint8) * np.array(mean255, dtype=np.uint8).reshape(1, 1, 3))
        out = []
        for image in images:
            if image is None:
                image = background
            if self.image_size is not None:
                sz = self.image_size
                if isinstance(sz, int):          # transforms.Resize(int): shorter side -> sz, bilinear
                    w, h = image.size
                    nw, nh = (sz, max(1, int(sz * h / w))) if w <= h else (max(1, int(sz * w / h)), sz)
                    image = image.resize((nw, nh), resample=Image.BILINEAR)
                else:
                    image = image.resize((sz[1], sz[0]), resample=Image.BILINEAR)
            if self.args["dataset"].get("auto_adjust_image_brightness", False):
                px = np.asarray(image.convert("RGB"), dtype=np.float64)
                if px.sum() / (px.shape[0] * px.shape[1] * 255.0 * 3) <= 0.15:
                    image = ImageEnhance.Brightness(image).enhance(1.75)
            if self.args["dataset"].get("image_aspect_ratio", "pad") == "pad":
                w, h = image.size
                if w != h:
                    side = max(w, h)
                    sq = Image.new(image.mode, (side, side), mean255)
                    sq.paste(image, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
                    image = sq
            out.append(self.image_processor.preprocess(image, return_tensors="pt")["pixel_values"][0])
        return torch.stack(out, dim=0)

    @torch.no_grad()
    def step(self, proprio, images, text_embeds):
        """proprio [1, 10]; images: [ext_{t-1}, right_wrist_{t-1}, left_wrist_{t-1}, ext_t, right_wrist_t, left_wrist_t] (PIL or
        None); text_embeds [1, L, lang_token_dim].  Returns the action chunk [1, horizon, 10] fp32 (gripper back in 0..255)."""
        device, dtype = self.device, self.dtype
        image_tensor = self.preprocess_images(images).to(device, dtype=dtype)
        image_embeds = self.vision_model(image_tensor).detach()
        image_embeds = image_embeds.reshape(-1, self.vision_model.hidden_size).unsqueeze(0)
        joints = proprio.to(device).unsqueeze(0)
        states, mask = self._format_joint_to_state(joints)
        states, mask = states.to(device, dtype=dtype), mask.to(device, dtype=dtype)
        states = states[:, -1:, :]
        ctrl_freqs = torch.tensor([self.control_frequency]).to(device)
        text_embeds = text_embeds.to(device, dtype=dtype)
        trajectory = self.policy.predict_action(
            lang_tokens=text_embeds, lang_attn_mask=torch.ones(text_embeds.shape[:2], dtype=torch.bool, device=text_embeds.device),
            img_tokens=image_embeds, state_tokens=states, action_mask=mask.unsqueeze(1), ctrl_freqs=ctrl_freqs)
        return self._unformat_action_to_joint(trajectory).to(torch.float32)
