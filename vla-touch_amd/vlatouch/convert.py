"""Episode files: raw recording folder -> `episode_*.h5`, and the labelled (controller-dataset) episode writer.

Mirrors the reference's data-preparation scripts on top of vlatouch.h5lite (no h5py / cv2 here):
  * `convert_episode_to_hdf5(episode_path, output_path)` / `convert_dataset_to_hdf5(input_dir, output_dir)`
        VLA/data/franka_data/4_convert_to_hdf5.py:18-199 — every top-level `*.npy` becomes a dataset named after the file, a
        `*.pt` instruction embedding becomes `instruct_embeddings` (float32), every sub-folder becomes a GROUP holding its images
        (sorted by the number in `rgb_<n>.jpg`, decoded to RGB) as ONE array dataset named like the folder plus one dataset per
        `*.npy` inside; all datasets LZF-compressed.  Images of unequal size are area-resized to the smallest (the reference's
        `cv2.INTER_AREA` fallback; here PIL's BOX filter) and the group gets `resized`-style bookkeeping in the returned report
        (h5lite writes no attributes).  JPEG decoding is PIL's (libjpeg) instead of cv2.imread's: third-party decoders, unpinned.
  * `write_labelled_episode(input_path, output_path, vla_action, camera1_resized, camera2_resized)`
        VLA/data/create_controller_dataset_episode.py:161-213 — copies every dataset / group of the input episode and adds
        `vla_action [N, chunk, 10] float32` and the two `camera*_resized [N, 384, 384, 3] uint8` arrays the controller dataset
        reads (residual_controller/controller_dataset.py:101-170).
"""
from __future__ import annotations

import os
import re
from typing import Dict, List

import numpy as np

from . import h5lite


def get_file_number(filename: str) -> int:
    m = re.search(r'rgb_(\d+)\.jpg', filename)
    return int(m.group(1)) if m else 0


def _imread_rgb(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def episode_tree(episode_path: str, report: Dict[str, object] = None) -> Dict[str, object]:
    """The {name: array | {name: array}} tree `convert_episode_to_hdf5` stores."""
    from PIL import Image
    tree: Dict[str, object] = {}
    for item_name in os.listdir(episode_path):
        item_path = os.path.join(episode_path, item_name)
        if os.path.isfile(item_path) and item_path.endswith('.npy'):
            tree[os.path.splitext(item_name)[0]] = np.load(item_path, allow_pickle=True)
            continue
        if os.path.isfile(item_path) and item_path.endswith('.pt'):
            import torch
            tree["instruct_embeddings"] = torch.load(item_path, map_location="cpu").to(torch.float32).numpy()
            continue
        if not os.path.isdir(item_path):
            continue
        group: Dict[str, np.ndarray] = {}
        files = [f for f in os.listdir(item_path) if os.path.isfile(os.path.join(item_path, f))]
        image_files = sorted((f for f in files if f.lower().endswith(('.jpg', '.jpeg', '.png'))), key=get_file_number)
        if image_files:
            imgs = [_imread_rgb(os.path.join(item_path, f)) for f in image_files]
            if len({im.shape for im in imgs}) > 1:           # unequal sizes: resize everything to the smallest (INTER_AREA)
                h, w = min(im.shape[0] for im in imgs), min(im.shape[1] for im in imgs)
                imgs = [np.asarray(Image.fromarray(im).resize((w, h), Image.BOX)) for im in imgs]
                if report is not None:
                    report[item_name] = {"resized": True, "target_width": w, "target_height": h}
            group[item_name] = np.stack(imgs)
        for f in sorted((f for f in files if f.endswith('.npy')), key=get_file_number):
            group[os.path.splitext(f)[0]] = np.load(os.path.join(item_path, f))
        tree[item_name] = group
    return tree


def convert_episode_to_hdf5(episode_path: str, output_path: str) -> Dict[str, object]:
    os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
    report: Dict[str, object] = {}
    h5lite.write_file(output_path, episode_tree(episode_path, report), compression="lzf")
    return report


def convert_dataset_to_hdf5(input_dir: str, output_dir: str) -> List[str]:
    """Every sub-folder of `input_dir` is one episode -> `output_dir/<episode>.h5` (4_convert_to_hdf5.py:170-199)."""
    os.makedirs(output_dir, exist_ok=True)
    out = []
    for name in sorted(os.listdir(input_dir)):
        ep = os.path.join(input_dir, name)
        if os.path.isdir(ep):
            dst = os.path.join(output_dir, f"{name}.h5")
            convert_episode_to_hdf5(ep, dst)
            out.append(dst)
    return out


def read_tree(path: str) -> Dict[str, object]:
    """An episode file as a nested dict of arrays (groups -> dicts)."""
    def walk(g):
        return {k: (walk(v) if isinstance(v, h5lite.Group) else v[...]) for k, v in g.items()}
    with h5lite.File(path) as f:
        return walk(f)


def write_labelled_episode(input_path: str, output_path: str, vla_action: np.ndarray, camera1_resized: np.ndarray, camera2_resized: np.ndarray) -> None:
    tree = read_tree(input_path)
    n = len(np.asarray(tree["ee_poses"]))
    vla_action = np.asarray(vla_action, dtype=np.float32)
    if vla_action.ndim != 3 or vla_action.shape[0] != n or vla_action.shape[2] != 10:
        raise ValueError(f"vla_action must be [N={n}, chunk, 10], got {vla_action.shape}")
    for nm, cam in (("camera1_resized", camera1_resized), ("camera2_resized", camera2_resized)):
        cam = np.asarray(cam)
        if cam.dtype != np.uint8 or cam.ndim != 4 or cam.shape[0] != n or cam.shape[-1] != 3:
            raise ValueError(f"{nm} must be uint8 [N={n}, H, W, 3], got {cam.dtype} {cam.shape}")
        tree[nm] = cam
    tree["vla_action"] = vla_action
    os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
    h5lite.write_file(output_path, tree, compression="lzf")
