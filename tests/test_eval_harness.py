"""the offline evaluation harness (vlatouch/eval.py) against goldens captured from the reference's own
ControllerDataset (run on two synthetic episodes through an h5py stand-in) and the evaluator's metric formulas."""
import glob
import os

import numpy as np
import pytest
import torch

from tests import cases

torch.set_grad_enabled(False)
EP = sorted(glob.glob(os.path.join(cases.GOLDEN, "episodes", "episode_*.npz")))


def G():
    return np.load(os.path.join(cases.GOLDEN, "g10_episode_eval.npz"))


def test_windows_samples_and_stats_match_reference_dataset():
    from vlatouch import eval as ev
    g = G()
    paths = ev.find_episodes(os.path.join(cases.GOLDEN, "episodes"))
    assert [os.path.basename(p) for p in paths] == ["episode_1.npz", "episode_2.npz"]
    eps = [ev.load_episode(p) for p in paths]
    index = [(e, s) for e, ep in enumerate(eps) for s in ev.episode_windows(ep, 2, 16, 1)]
    assert np.array_equal(np.array(index), g["episode_indices"])
    assert index[0][1] == 4                                   # first-motion trim: frames 0-3 are static
    st = ev.normalization_stats(eps)
    for k, v in st.items():
        assert np.abs(v - g["stats_" + k]).max() < 1e-9, k
    for idx in (0, 3, len(index) - 1):
        e, s = index[idx]
        smp = ev.make_sample(eps[e], s, 2, 16)
        for k in ("states", "vla_actions", "expert_actions", "forces", "disps"):
            assert np.abs(smp[k].numpy() - g[f"s{idx}_{k}"]).max() < 1e-6, (idx, k)
        sums = np.array([float(smp["images_cam1"].sum()), float(smp["images_cam2"].sum())])
        assert np.abs(sums - g[f"s{idx}_img1_sum"]).max() < 1e-2
    # the gripper /255 convention: expert (and the horizon part of `states`) rescaled, context states are not
    smp = ev.make_sample(eps[0], index[3][1], 2, 16)
    raw = ev.converted_ee_pose_with_gripper(eps[0])[index[3][1]:index[3][1] + 18]
    assert np.allclose(smp["states"][:2, -1].numpy(), raw[:2, -1], atol=1e-4) and np.allclose(smp["states"][2:, -1].numpy(), raw[2:, -1] / 255, atol=1e-6)
    assert ev.episode_windows({"ee_poses": np.zeros((30, 7))}, 2, 16) == []      # no motion -> episode skipped


def test_quaternion_to_6d_is_first_two_rotation_columns():
    from scipy.spatial.transform import Rotation as R
    from vlatouch.eval import quaternion_to_ortho6d
    q = np.random.default_rng(0).normal(size=(20, 4))
    six = quaternion_to_ortho6d(q)
    m = R.from_quat(q / np.linalg.norm(q, axis=1, keepdims=True)).as_matrix()
    assert np.abs(six - m[:, :, :2].transpose(0, 2, 1).reshape(20, 6)).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_evaluator_metrics_golden(prec):
    from residual_controller.bridge_controller import DiffusionController
    from vlatouch import eval as ev, synth
    g = G()
    eps = [ev.load_episode(p) for p in EP]
    ctrl = cases.build_controller(DiffusionController, precision=prec)
    ctrl.stats = {k: torch.as_tensor(v, dtype=torch.float32).cuda() for k, v in ev.normalization_stats(eps).items()}
    noises = {bi: cases.T(synth.inputs_rng(900 + bi).standard_normal((10, n, 16, 10), dtype=np.float32)) for bi, n in ((0, 32), (1, 12))}
    res = ev.evaluate(ctrl, eps, horizon=16, chosen=[0, 5, 40], noises=noises)
    bm = g["batch_metrics"]
    want_err = [bm[0, 0], bm[0, 0], bm[1, 0]]
    want_vla = [bm[0, 1], bm[0, 1], bm[1, 1]]
    tol = 2e-4 if prec == "fp32" else 2e-2
    assert np.allclose(res["test_vla_errors"], want_vla, rtol=1e-5)
    assert np.allclose(res["test_errors"], want_err, rtol=tol), (res["test_errors"], want_err)
    avg, avg_v = np.mean(want_err), np.mean(want_vla)
    assert abs(res["improvement"] - (1 - avg / avg_v) * 100) < (0.05 if prec == "fp32" else 2.0)
    # random selection path: deterministic under a seed, batch-level predict
    r2 = ev.evaluate(ctrl, eps, num_samples=2, horizon=16, seed=3)
    assert len(r2["test_errors"]) == 2 and r2["chosen"] == ev.evaluate(ctrl, eps, num_samples=2, horizon=16, seed=3)["chosen"]
    pred, expert, vla = ev.refine_batch(ctrl, next(ev.batches(eps, 32, 2, 16)), 2, noises[0])
    assert float((pred.cpu() - torch.from_numpy(g["pred_b0"])).abs().max()) < (2e-4 if prec == "fp32" else 3e-2)


@pytest.mark.gpu
def test_refine_episode_is_the_batch_path_over_one_episode():
    """vlatouch.eval.refine_episode (SURVEY §8a-10's entry point): every window of one episode, in loader order, equals the
    batch-by-batch refine_batch calls, and its metrics are the evaluator's formulas over the whole episode."""
    from residual_controller.bridge_controller import DiffusionController
    from vlatouch import eval as ev, synth
    eps = [ev.load_episode(p) for p in EP]
    ctrl = cases.build_controller(DiffusionController, precision="fp32")
    ctrl.stats = {k: torch.as_tensor(v, dtype=torch.float32).cuda() for k, v in ev.normalization_stats(eps).items()}
    ep = eps[0]
    nwin = len(ev.episode_windows(ep, 2, 16, 1))
    z = cases.T(synth.inputs_rng(77).standard_normal((10, nwin, 16, 10), dtype=np.float32)).cuda()
    pred, met = ev.refine_episode(ctrl, ep, batch_size=8, horizon=16, noise=z)
    assert pred.shape == (nwin, 16, 10)
    parts, exps, vlas, done = [], [], [], 0
    for b in ev.batches([ep], 8, 2, 16):
        n = b["states"].shape[0]
        p, e, v = ev.refine_batch(ctrl, b, 2, z[:, done:done + n].contiguous())
        parts.append(p); exps.append(e); vlas.append(v)
        done += n
    assert done == nwin and torch.equal(pred, torch.cat(parts))
    err = torch.mean((torch.cat(parts) - torch.cat(exps)) ** 2).item()
    verr = torch.mean((torch.cat(vlas) - torch.cat(exps)) ** 2).item()
    assert abs(met["error"] - err) < 1e-9 and abs(met["vla_error"] - verr) < 1e-9
    assert abs(met["improvement"] - (1 - err / verr) * 100) < 1e-6


def test_controller_dataset_and_data_module_mirrors():
    """residual_controller.controller_dataset.ControllerDataset / ControllerDataModule over the h5 episode fixtures: the index mapping,
    samples and statistics the REFERENCE class produced on the same two episodes (g10), and the file-level train / val split."""
    from residual_controller.controller_dataset import ControllerDataModule, ControllerDataset
    g = G()
    root = os.path.join(cases.GOLDEN, "episodes_h5")
    files = [os.path.join(root, f) for f in ("episode_1.h5", "episode_2.h5")]
    ds = ControllerDataset(root, file_paths=files, context_frames=2, horizon=16, use_images=True)
    assert len(ds) == len(g["episode_indices"]) and np.array_equal(np.array(ds.episode_indices), g["episode_indices"])
    for k, v in ds.stats.items():
        assert np.abs(v - g["stats_" + k]).max() < 1e-9, k
    for idx in (0, 3, len(ds) - 1):
        smp = ds[idx]
        for k in ("states", "vla_actions", "expert_actions", "forces", "disps"):
            assert np.abs(smp[k].numpy() - g[f"s{idx}_{k}"]).max() < 1e-6, (idx, k)
        assert smp["images_cam1"].shape == (2, 28, 28, 3) and float(smp["images_cam1"].max()) <= 1.0
    np.random.seed(0)
    import tempfile, shutil
    with tempfile.TemporaryDirectory() as d:           # only the two episode files (storage_forms.h5 in the fixture dir is not an episode)
        for f in files:
            shutil.copy(f, d)
        dm = ControllerDataModule(d, batch_size=8, num_workers=0, context_frames=2, horizon=16, use_images=True, val_ratio=0.1)
        assert len(dm.train_dataset.file_paths) == 1 and len(dm.val_dataset.file_paths) == 1
        assert set(dm.train_dataset.file_paths) | set(dm.val_dataset.file_paths) == {os.path.join(d, os.path.basename(f)) for f in files}
        tb = list(dm.train_dataloader())
        assert len(tb) == len(dm.train_dataset) // 8 and all(b["states"].shape == (8, 18, 10) for b in tb)      # drop_last
        vb = list(dm.val_dataloader())
        assert sum(b["states"].shape[0] for b in vb) == len(dm.val_dataset)
        assert vb[0]["images_cam1"].shape[1:] == (2, 28, 28, 3) and vb[0]["vla_actions"].shape[1:] == (16, 10)
        one = ControllerDataset(d, file_paths=dm.train_dataset.file_paths, context_frames=2, horizon=16)
        for k, v in dm.stats.items():
            assert np.array_equal(v, one.stats[k])
