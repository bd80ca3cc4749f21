"""Action (de)normalisation — mirror of the reference's
VLA/residual_controller/controller_dataset.py:303-346 (`normalize_actions`) and :349-384
(`denormalize_actions`): padded (x1.4) min-max range mapped to [-1, 1]; the <1e-6 range guard exists in
`normalize_actions` only; unknown `action_type` raises ValueError.  The arithmetic runs in the HIP kernel
behind `vt_action_normalize` (include/vlatouch.h); CPU tensors are staged through the GPU and returned on CPU.
`ControllerDataset` / `ControllerDataModule` (:30-236, :386-491) feed the trainer mirrors (bridge_train.py, lstm_train.py) from the
reference's `episode_*.h5` files; their window / statistics arithmetic is `vlatouch.eval`'s (pinned by g10).
"""
from __future__ import annotations

import torch

from vlatouch import _lib as L
from vlatouch.engine import action_normalize


def _select(stats, action_type):
    if action_type == "expert":
        return stats["action_mins"], stats["action_maxs"]
    if action_type == "vla":
        return stats["vla_mins"], stats["vla_maxs"]
    raise ValueError(f"Unknown action_type: {action_type}")


def _run(x, stats, action_type, padding_factor, denorm):
    mins, maxs = _select(stats, action_type)
    x = torch.as_tensor(x)
    src_device = x.device
    dev = src_device if src_device.type == "cuda" else L.require_gpu("cuda")
    out = action_normalize(x.to(dev), torch.as_tensor(mins), torch.as_tensor(maxs), denorm=denorm, padding_factor=padding_factor)
    return out.to(src_device)


def normalize_actions(actions, stats, action_type="expert", padding_factor=1.4):
    return _run(actions, stats, action_type, padding_factor, False)


def denormalize_actions(normalized_actions, stats, action_type="expert", padding_factor=1.4):
    return _run(normalized_actions, stats, action_type, padding_factor, True)


# ------------------------------------------------------------------------------------------------ dataset / data module
class ControllerDataset(torch.utils.data.Dataset):
    """Windows of the VLA-augmented episodes (controller_dataset.py:30-236): same constructor, `file_paths`, `create_index_mapping`,
    `episode_indices`, `__len__`, `__getitem__` keys ('states', 'vla_actions', 'expert_actions', 'forces', 'disps', 'images_cam1',
    'images_cam2') and `get_normalization_stats`.  The arithmetic is `vlatouch.eval`'s restatement (pinned to the reference class by
    g10); episodes are read with the product's own HDF5 / LZF reader (`vlatouch.h5lite`) and kept in a small per-process cache, since one
    window touches a few rows of every dataset of its file."""

    def __init__(self, data_dir, file_paths=None, context_frames=2, horizon=8, use_images=False, image_size=384, stride=1, cache_episodes=8):
        from vlatouch import eval as ev
        self.data_dir, self.context_frames, self.horizon = data_dir, context_frames, horizon
        self.use_images, self.image_size, self.stride = use_images, image_size, stride
        if file_paths is None:
            self.file_paths = [p for p in ev.find_episodes(data_dir) if p.endswith(".h5")]
        else:
            self.file_paths = list(file_paths)
        self._cache, self._cache_max = {}, cache_episodes
        self.create_index_mapping()
        self.stats = self.get_normalization_stats()

    def _episode(self, path):
        from vlatouch import eval as ev
        ep = self._cache.get(path)
        if ep is None:
            if len(self._cache) >= self._cache_max:
                self._cache.pop(next(iter(self._cache)))
            ep = self._cache[path] = ev.load_episode(path, images=self.use_images)     # camera streams only when windows need them
        return ep

    def _light(self):
        """Pose / action arrays of every file, one episode alive at a time, outside the window cache (the statistics and the index
        mapping walk the whole dataset once; the reference reads only these arrays per file for them)."""
        from vlatouch import eval as ev
        for path in self.file_paths:
            yield path, ev.load_episode(path, images=False)

    def create_index_mapping(self):
        from vlatouch import eval as ev
        self.episode_indices, self.total_samples = [], 0
        light = []
        for file_idx, (path, ep) in enumerate(self._light()):
            light.append({k: ep[k] for k in ("ee_poses", "gripper_pos", "vla_action") if k in ep})
            starts = ev.episode_windows(ep, self.context_frames, self.horizon, self.stride)
            if not starts:
                print(f"Warning: No movement detected in file {path}. Skipping.")
                continue
            self.episode_indices.extend((file_idx, s) for s in starts)
            self.total_samples += len(starts)
        self._stats = ev.normalization_stats(light) if light else None       # same pass: the files are read once

    def __len__(self):
        return self.total_samples

    def __getitem__(self, idx):
        from vlatouch import eval as ev
        file_idx, start = self.episode_indices[idx]
        return ev.make_sample(self._episode(self.file_paths[file_idx]), start, self.context_frames, self.horizon, self.use_images)

    def get_normalization_stats(self):
        from vlatouch import eval as ev
        if getattr(self, "_stats", None) is None:
            self._stats = ev.normalization_stats(ep for _, ep in self._light())
        return self._stats


class ControllerDataModule:
    """Train / validation split over episode FILES and the two loaders (controller_dataset.py:386-491): same constructor, `setup`,
    `train_dataset`, `val_dataset`, `stats` (of the training files), `train_dataloader` (shuffled, drop_last), `val_dataloader`."""

    def __init__(self, data_dir, batch_size=32, num_workers=4, context_frames=2, horizon=8, use_images=True, image_size=384, val_ratio=0.1, stride=1):
        self.data_dir, self.batch_size, self.num_workers = data_dir, batch_size, num_workers
        self.context_frames, self.horizon, self.use_images = context_frames, horizon, use_images
        self.image_size, self.val_ratio, self.stride = image_size, val_ratio, stride
        self.setup()

    def setup(self):
        import numpy as np
        from vlatouch import eval as ev
        print(f"loading dataset from {self.data_dir} ..")
        file_paths = [p for p in ev.find_episodes(self.data_dir) if p.endswith(".h5")]
        num_val = max(1, int(len(file_paths) * self.val_ratio))
        val_indices = np.random.choice(len(file_paths), num_val, replace=False)
        train_files = [file_paths[i] for i in range(len(file_paths)) if i not in val_indices]
        val_files = [file_paths[i] for i in val_indices]
        kw = dict(data_dir=self.data_dir, context_frames=self.context_frames, horizon=self.horizon, use_images=self.use_images,
                  image_size=self.image_size, stride=self.stride)
        self.train_dataset = ControllerDataset(file_paths=train_files, **kw)
        self.val_dataset = ControllerDataset(file_paths=val_files, **kw)
        self.stats = self.train_dataset.stats                       # computed once, in the dataset's constructor

    def train_dataloader(self):
        return torch.utils.data.DataLoader(self.train_dataset, batch_size=self.batch_size, shuffle=True, num_workers=self.num_workers,
                                           pin_memory=torch.cuda.is_available(), drop_last=True)

    def val_dataloader(self):
        return torch.utils.data.DataLoader(self.val_dataset, batch_size=self.batch_size, shuffle=False, num_workers=self.num_workers,
                                           pin_memory=torch.cuda.is_available(), drop_last=False)
