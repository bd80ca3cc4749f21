"""Offline evaluation harness of the refinement path: episode windowing, the `/255` gripper convention,
normalisation statistics, batch-level `predict` calls and the three metrics of the reference's evaluator.

Restates (host-side data plumbing, numpy):
  * `converted_ee_pose_with_gripper` / quaternion -> 6-D rotation     VLA/scripts/utils_eef.py:80-100, VLA/docs/test_6drot.py:5-116
  * `ControllerDataset.create_index_mapping / __getitem__ / get_normalization_stats`
                                                                      VLA/residual_controller/controller_dataset.py:71-236
  * `test_diffusion_controller` metrics                               VLA/residual_controller/bridge_test.py:111-197
The refinement itself is `controller.predict` (HIP engines).  Episodes are read from the reference's `episode_*.h5` files (LZF-
compressed HDF5 as written by data/franka_data/4_convert_to_hdf5.py / create_controller_dataset_episode.py, parsed by
vlatouch/h5lite.py) or from NPZ files with the same key names:
    ee_poses [N,7] (xyz + quaternion xyzw), gripper_pos [N], vla_action [N,64,10],
    gelsight_force/forces [N,3], gelsight_force/displacement [N,63,2] (optional), camera1_resized / camera2_resized [N,res,res,3] uint8
"""
from __future__ import annotations

import fnmatch
import os
import random
import re
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

EPS_MOTION = 1e-2     # controller_dataset.py:80


def quaternion_to_ortho6d(quat: np.ndarray) -> np.ndarray:
    """[N,4] xyzw quaternions -> [N,6] = first two columns of the rotation matrix (test_6drot.py:75-81,109-116; the
    reference goes quaternion -> euler -> matrix through scipy, which is the same rotation)."""
    q = np.asarray(quat, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    c0 = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)], axis=1)
    c1 = np.stack([2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)], axis=1)
    return np.concatenate([c0, c1], axis=1)


def converted_ee_pose_with_gripper(ep: Dict[str, np.ndarray]) -> np.ndarray:
    """[ee_pos(3) | 6-D rotation | gripper] -> [N,10]  (utils_eef.py:80-90)."""
    poses = np.asarray(ep["ee_poses"], dtype=np.float64)
    grip = np.asarray(ep["gripper_pos"], dtype=np.float64).reshape(-1, 1)
    return np.concatenate([poses[:, :3], quaternion_to_ortho6d(poses[:, 3:]), grip], axis=-1)


def natural_sort_filenames(files: Sequence[str]) -> List[str]:
    def num(f):
        m = re.search(r"episode_(\d+)", f)
        return int(m.group(1)) if m else 0
    return sorted(files, key=num)


def load_episode(path: str, images: bool = True) -> Dict[str, np.ndarray]:
    """images=False skips the two camera streams (442 KB per frame each once decoded): statistics, index mapping and image-free
    training only touch the pose / action / force arrays (controller_dataset.py:172-236 reads just those per file)."""
    cams = ("camera1_resized", "camera2_resized") if images else ()
    if path.endswith(".npz"):
        z = np.load(path)
        ep = {k: z[k] for k in z.files if images or k not in ("camera1_resized", "camera2_resized", "camera1_images", "camera2_images")}
    else:
        from . import h5lite       # the product's own reader of the reference's LZF-compressed episode files (no h5py needed)
        ep = {}
        with h5lite.File(path, "r") as f:
            for k in ("ee_poses", "gripper_pos", "vla_action") + cams:
                if k in f:
                    ep[k] = f[k][:]
            ep["gelsight_force/forces"] = f["gelsight_force"]["forces"][:]
            if "displacement" in f["gelsight_force"]:
                ep["gelsight_force/displacement"] = f["gelsight_force"]["displacement"][:]
    if "gelsight_force/forces" not in ep and "forces" in ep:
        ep["gelsight_force/forces"] = ep["forces"]
    return ep


def find_episodes(data_dir: str) -> List[str]:
    out = []
    for root, _, files in os.walk(data_dir):
        names = fnmatch.filter(files, "*.h5") + fnmatch.filter(files, "*.npz")
        for fn in natural_sort_filenames(names):
            out.append(os.path.join(root, fn))
    return out


def episode_windows(ep: Dict[str, np.ndarray], context_frames: int = 2, horizon: int = 8, stride: int = 1) -> List[int]:
    """Valid window starts: from the first frame whose pose moved by > 1e-2 from frame 0 (any of the 7 pose components) up to
    the last start that still leaves context_frames + horizon frames (controller_dataset.py:78-96)."""
    q = np.asarray(ep["ee_poses"])
    moved = np.where(np.any(np.abs(q - q[0:1]) > EPS_MOTION, axis=1))[0]
    if len(moved) == 0:
        return []
    return list(range(int(moved[0]), q.shape[0] - (context_frames + horizon - 1), stride))


def make_sample(ep: Dict[str, np.ndarray], start: int, context_frames: int = 2, horizon: int = 8, use_images: bool = True) -> Dict[str, torch.Tensor]:
    """One window (controller_dataset.py:101-170): states [cf+h,10] (the horizon part has gripper/255 — the reference's
    in-place view semantics), vla_actions [h,10] (gripper/255), expert_actions [h,10], forces [cf+h,3], images [cf,H,W,3]/255."""
    cf, h = context_frames, horizon
    qpos = converted_ee_pose_with_gripper(ep)[start:start + cf + h].copy()
    qpos[cf:, -1] /= 255                          # `future_states` is a VIEW of qpos in the reference: states see it too
    vla = np.array(ep["vla_action"][start + cf][:h], dtype=np.float64)
    vla[:, -1] /= 255
    forces = np.asarray(ep["gelsight_force/forces"][start:start + cf + h])
    out = {"states": torch.as_tensor(qpos, dtype=torch.float32), "vla_actions": torch.as_tensor(vla, dtype=torch.float32),
           "expert_actions": torch.as_tensor(qpos[cf:], dtype=torch.float32), "forces": torch.as_tensor(forces, dtype=torch.float32)}
    if "gelsight_force/displacement" in ep:
        out["disps"] = torch.as_tensor(np.asarray(ep["gelsight_force/displacement"][start:start + cf + h]), dtype=torch.float32)
    if use_images:
        out["images_cam1"] = torch.as_tensor(np.asarray(ep["camera1_resized"][start:start + cf]), dtype=torch.float32) / 255.0
        out["images_cam2"] = torch.as_tensor(np.asarray(ep["camera2_resized"][start:start + cf]), dtype=torch.float32) / 255.0
    return out


def normalization_stats(episodes) -> Dict[str, np.ndarray]:
    """Per-dimension min/max of expert and VLA actions over whole episodes, gripper/255 (controller_dataset.py:172-236).  `episodes` is any
    iterable (a generator keeps one episode alive at a time)."""
    amin, amax = np.full(10, np.inf), np.full(10, -np.inf)
    vmin, vmax = np.full(10, np.inf), np.full(10, -np.inf)
    for ep in episodes:
        ex = converted_ee_pose_with_gripper(ep)
        ex[:, -1] /= 255
        va = np.array(ep["vla_action"], dtype=np.float64)
        va[:, :, -1] /= 255
        amin, amax = np.minimum(amin, ex.min(0)), np.maximum(amax, ex.max(0))
        vmin, vmax = np.minimum(vmin, va.min((0, 1))), np.maximum(vmax, va.max((0, 1)))
    ar, vr = amax - amin, vmax - vmin
    ar[ar < 1e-6] = 1.0
    vr[vr < 1e-6] = 1.0
    return {"action_mins": amin, "action_maxs": amax, "vla_mins": vmin, "vla_maxs": vmax, "action_range": ar, "vla_range": vr}


def collate(samples: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    return {k: torch.stack([s[k] for s in samples]) for k in samples[0]}


def batches(episodes: Sequence[Dict[str, np.ndarray]], batch_size: int = 32, context_frames: int = 2, horizon: int = 8, stride: int = 1):
    """Validation-loader order: windows of episode 0, 1, ... in order, fixed-size batches, last one partial (shuffle=False)."""
    index = [(e, s) for e, ep in enumerate(episodes) for s in episode_windows(ep, context_frames, horizon, stride)]
    for i in range(0, len(index), batch_size):
        yield collate([make_sample(episodes[e], s, context_frames, horizon) for e, s in index[i:i + batch_size]])


def refine_batch(controller, batch: Dict[str, torch.Tensor], context_frames: int = 2, noise: Optional[torch.Tensor] = None):
    """The reference's per-batch call (bridge_test.py:131-173): current state/force = last context frame, image = last context
    image, whole batch through `predict`.  Returns (predicted, expert, vla)."""
    cf = context_frames
    dev = controller.device
    states, forces = batch["states"].to(dev), batch["forces"].to(dev)
    pred = controller.predict(states[:, cf - 1], batch["vla_actions"].to(dev), batch["images_cam1"][:, -1].to(dev),
                              batch["images_cam2"][:, -1].to(dev), forces[:, cf - 1], **({"noise": noise} if noise is not None else {}))
    return pred, batch["expert_actions"].to(dev), batch["vla_actions"].to(dev)


def evaluate(controller, episodes: Sequence[Dict[str, np.ndarray]], num_samples: int = 10, batch_size: int = 32, horizon: int = 32,
             stride: int = 1, seed: Optional[int] = None, chosen: Optional[Sequence[int]] = None, noises: Optional[Dict[int, torch.Tensor]] = None):
    """bridge_test.py:111-197: pick `num_samples` window indices at random; for each, run `predict` on the WHOLE batch holding it
    and record the batch-level MSE(pred, expert) and MSE(vla, expert); report the averages and the improvement %."""
    cf = controller.model_args.get("context_frames", 2) if controller.model_args else 2
    bl = list(batches(episodes, batch_size, cf, horizon, stride))
    flat = [(i, j) for i, b in enumerate(bl) for j in range(b["states"].shape[0])]
    if chosen is None:
        rng = random.Random(seed)
        chosen = rng.sample(range(len(flat)), num_samples) if num_samples <= len(flat) else list(range(len(flat)))
    errs, vla_errs = [], []
    for idx in chosen:
        bi, _ = flat[idx]
        pred, expert, vla = refine_batch(controller, bl[bi], cf, None if noises is None else noises.get(bi))
        errs.append(torch.mean((pred - expert) ** 2).item())
        vla_errs.append(torch.mean((vla - expert) ** 2).item())
    avg, avg_vla = sum(errs) / len(errs), sum(vla_errs) / len(vla_errs)
    return {"avg_error": avg, "avg_vla_error": avg_vla, "improvement": (1.0 - avg / avg_vla) * 100 if avg_vla > 0 else 0,
            "test_errors": errs, "test_vla_errors": vla_errs, "chosen": list(chosen)}


def refine_episode(controller, episode: Dict[str, np.ndarray], batch_size: int = 32, horizon: int = 32, stride: int = 1,
                   noise: Optional[torch.Tensor] = None):
    """All windows of ONE episode through `predict`, in validation-loader order: returns the refined chunks
    [n_windows, horizon, 10], and the evaluator's three metrics over the whole episode — MSE(pred, expert), MSE(vla, expert),
    improvement %.  `noise` (optional) is [10, n_windows, horizon, 10], sliced per batch."""
    cf = controller.model_args.get("context_frames", 2) if controller.model_args else 2
    preds, experts, vlas = [], [], []
    done = 0
    for b in batches([episode], batch_size, cf, horizon, stride):
        n = b["states"].shape[0]
        z = None if noise is None else noise[:, done:done + n].contiguous()
        p, e, v = refine_batch(controller, b, cf, z)
        preds.append(p); experts.append(e); vlas.append(v)
        done += n
    if not preds:
        return torch.empty(0, horizon, 10), {"error": float("nan"), "vla_error": float("nan"), "improvement": 0.0}
    pred, expert, vla = torch.cat(preds), torch.cat(experts), torch.cat(vlas)
    err = torch.mean((pred - expert) ** 2).item()
    verr = torch.mean((vla - expert) ** 2).item()
    return pred, {"error": err, "vla_error": verr, "improvement": (1.0 - err / verr) * 100 if verr > 0 else 0.0}
