"""Controller training step on the device against golden vectors captured from the REFERENCE's own autograd run
(tools/make_golden_train.py: StochasticInterpolants.get_loss + backward + AdamW + EMA, two steps, bridge_model.py:183-258,
bridge_train.py:49-58, 312-334).  fp32 end to end; bars: losses 1e-5 relative, gradients / updated parameters / EMA shadows 1e-4 of the
tensor's norm (summaries: norm, projection on a seeded direction, first values), d loss / d obs_cond element-wise."""
import numpy as np
import pytest
import torch

from tests import cases
from tools.make_golden_train import train_inputs, direction

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def G():
    return np.load(f"{cases.GOLDEN}/g13_train.npz")


def summarize(name, a):
    v = a.detach().double().cpu().numpy()
    return np.concatenate([[np.sqrt((v * v).sum()), (v * direction(name, v.shape).astype(np.float64)).sum()], v.reshape(-1)[:4]])


def check(table, names, tensors, what, rel):
    worst = 0.0
    for i, k in enumerate(names):
        if k not in tensors:
            continue
        got, want = summarize(k, tensors[k]), table[i]
        scale = max(want[0], 1e-12)
        e = np.abs(got - want).max() / scale
        worst = max(worst, e)
        assert e < rel, (what, k, e, got[:3], want[:3])
    return worst


def test_training_step_matches_reference_autograd_adamw_ema():
    from vlatouch.train import SITrainer
    g = G()
    names = [str(n) for n in g["names"]]
    tr = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), lr=1e-4, weight_decay=1e-6, device="cuda:0")
    for step in (1, 2):
        inp = train_inputs(step)
        loss, info = tr.get_loss(inp["obs_in"], inp["vla_n"], inp["expert_n"], inp["t"], inp["z"])
        want = g[f"s{step}_loss"]
        got = np.array([loss, info["v_loss"], info["s_loss"], info["b_loss"]])
        assert np.abs(got - want).max() < 1e-5 * np.abs(want).max(), (got, want)
        e = float(np.abs(tr.last_dcond.cpu().numpy() - g[f"s{step}_dcond"]).max())
        assert e < 1e-4 * float(np.abs(g[f"s{step}_dcond"]).max()), e
        grads = dict(tr.net_grads())
        grads.update({"state_encoder." + k: v for k, v in tr.mlp.grads().items()})
        assert set(names) == set(grads), set(names) ^ set(grads)
        wg = check(g[f"s{step}_grad"], names, grads, "grad", 1e-4)
        tr.optimizer_step()
        params = dict(tr.net_state_dict())
        params.update({"state_encoder." + k: v for k, v in tr.mlp.state_dict().items()})
        wp = check(g[f"s{step}_param"], names, params, "param", 1e-6)
        we = check(g[f"s{step}_ema"], names[:len(g[f"s{step}_ema"])], dict(tr.ema_state_dict()), "ema", 1e-6)
        print(f"[train step {step}] loss {loss:.6f}; worst relative error: grads {wg:.2e}, params {wp:.2e}, ema {we:.2e}")


INTERPOLANT_KINDS = ["power3", "power4", "reverse_power3", "reverse_power4", "gaussian_encode_decode", "reverse_linear"]


@pytest.mark.parametrize("kind", INTERPOLANT_KINDS)
def test_training_step_nonlinear_interpolants_match_reference_autograd(kind):
    """Round 6: every `interpolant_type` of the reference (bridge_model.py:103-147 `interpolant`, :149-181 `interpolant_dev`) in the device
    training step — q_sample + the three loss targets (vt_si_qsample_ex) -> losses, d loss / d obs_cond and all parameter gradients against ONE get_loss +
    backward of the reference itself per interpolant (tools/make_golden_train.py --interpolants -> g13_train_interpolants.npz), each paired with one of the three
    gamma schedules; t includes both clipped ends and both sides of the piecewise interpolants' indicator (0.5, 0.5 + 1 ulp)."""
    from vlatouch.train import SITrainer
    g = np.load(f"{cases.GOLDEN}/g13_train_interpolants.npz")
    names = [str(n) for n in g["names"]]
    gamma = str(g[f"{kind}_gamma"])
    tr = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), gamma_type=gamma, interpolant_type=kind, lr=1e-4, weight_decay=1e-6, device="cuda:0")
    inp = train_inputs(1)
    inp["t"] = cases.T(g["t"])
    loss, info = tr.get_loss(inp["obs_in"], inp["vla_n"], inp["expert_n"], inp["t"], inp["z"])
    want = g[f"{kind}_loss"]
    got = np.array([loss, info["v_loss"], info["s_loss"], info["b_loss"]])
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max(), (kind, got, want)
    e = float(np.abs(tr.last_dcond.cpu().numpy() - g[f"{kind}_dcond"]).max())
    assert e < 1e-4 * float(np.abs(g[f"{kind}_dcond"]).max()), (kind, e)
    grads = dict(tr.net_grads())
    grads.update({"state_encoder." + k: v for k, v in tr.mlp.grads().items()})
    wg = check(g[f"{kind}_grad"], names, grads, "grad", 1e-4)
    print(f"[train {kind} / {gamma}] loss {loss:.6f}; worst relative gradient error {wg:.2e}")


def test_training_step_rejects_unknown_interpolant():
    from vlatouch.train import SITrainer
    with pytest.raises(NotImplementedError):
        SITrainer(cases.si_net_sd(""), None, interpolant_type="cubic", device="cuda:0")


def test_training_primitives_against_torch_cpu():
    """The data-movement pieces (transposed im2col, flipped weights, zero stuffing, column sums) and the activation derivatives."""
    from vlatouch import train as tt
    dev = "cuda:0"
    rng = np.random.default_rng(3)
    x = cases.T(rng.standard_normal((4, 8, 16), dtype=np.float32)).to(dev)
    xt = tt.im2col_t(x, 4, 3, 2, -1).cpu()                       # k3 s2 p1: Tout 4
    xp = torch.nn.functional.pad(x.cpu(), (0, 0, 1, 1))
    ref = torch.stack([xp[:, 2 * t:2 * t + 3, :].reshape(4, 48) for t in range(4)], dim=1).reshape(16, 48).t()
    assert torch.equal(xt, ref)
    w = cases.T(rng.standard_normal((6, 3 * 16), dtype=np.float32)).to(dev)
    wf = tt.wflip(w, 6, 3, 16).cpu().reshape(16, 3, 6)
    assert torch.equal(wf, w.cpu().reshape(6, 3, 16).flip(1).permute(2, 1, 0))
    z = tt.zero_stuff(x).cpu()
    assert torch.equal(z[:, 0::2], x.cpu()) and float(z[:, 1::2].abs().max()) == 0.0
    m = cases.T(rng.standard_normal((37, 50), dtype=np.float32)).to(dev)
    assert torch.equal(tt.transpose(m).cpu(), m.cpu().t()) and torch.allclose(tt.colsum(m).cpu(), m.cpu().sum(0), atol=1e-5)
    v = cases.T(np.linspace(-6, 25, 4096, dtype=np.float32)).to(dev)
    dy = torch.ones_like(v)
    vr = v.cpu().double().requires_grad_(True)
    with torch.enable_grad():
        torch.nn.functional.mish(vr).sum().backward()
    assert float((tt.mish(v, dy).cpu().double() - vr.grad).abs().max()) < 2e-6
    vg = v.cpu().double().requires_grad_(True)
    with torch.enable_grad():
        torch.nn.functional.gelu(vg).sum().backward()
    assert float((tt.gelu(v, dy).cpu().double() - vg.grad).abs().max()) < 2e-6


def _batch(seed, B=8, T=16):
    rng = np.random.default_rng(seed)
    f = lambda *s: cases.T(rng.standard_normal(s, dtype=np.float32))
    return {"states": f(B, 2 + T, 10), "forces": f(B, 2 + T, 3), "vla_actions": cases.T(rng.random((B, T, 10), dtype=np.float32)),
            "expert_actions": cases.T(rng.random((B, T, 10), dtype=np.float32)),
            "images_cam1": cases.T(rng.integers(0, 256, (B, 1, 224, 224, 3), dtype=np.uint8)),
            "images_cam2": cases.T(rng.integers(0, 256, (B, 1, 224, 224, 3), dtype=np.uint8))}


class _DataModule:
    def __init__(self, stats):
        self.stats = stats


def test_trainer_mirror_steps_sync_and_checkpoint_round_trip(tmp_path):
    """DiffusionControllerTrainer (bridge_train.py:27): fixed batch, fixed draws -> the loss falls; the trained tensors land in the
    controller objects; save -> load restores parameters, EMA and num_updates; `diffusion_model.get_loss` agrees with the trainer."""
    from vlatouch import synth
    from residual_controller.bridge_controller import DiffusionController
    from residual_controller.bridge_train import DiffusionControllerTrainer
    dev = "cuda:0"
    ctrl = synth.build_controller(DiffusionController, "fp32", device=dev)
    stats = {k: v.cpu().numpy() for k, v in synth.unit_stats().items()}
    tr = DiffusionControllerTrainer(ctrl, _DataModule(stats), learning_rate=1e-3, checkpoint_dir=str(tmp_path / "ckpt"), device=dev)
    batch = _batch(5)
    g = torch.Generator().manual_seed(0)
    t, z = torch.rand(8, generator=g), torch.randn(8, 16, 10, generator=g)
    before = {k: v.clone() for k, v in ctrl.diffusion_model.net.state_dict().items()}
    l0, _ = tr.eval_step(batch, t, z)
    bd = tr._prepare_batch_for_diffusion(batch)
    obs_cond = ctrl.encode_observation(batch["states"][:, 1], batch["images_cam1"][:, -1], batch["images_cam2"][:, -1], batch["forces"][:, 1])
    lm, info = ctrl.diffusion_model.get_loss({"obs_cond": obs_cond, "vla_act": bd["vla_act"], "expert_act": bd["expert_act"]}, dev, t=t, z=z,
                                             backward=False)
    assert abs(float(lm) - l0) < 1e-4 * max(1.0, abs(l0)), (float(lm), l0)
    losses = [tr.train_step(batch, t, z)[0] for _ in range(12)]
    l1, _ = tr.eval_step(batch, t, z)
    assert np.isfinite(losses).all() and l1 < l0, (l0, losses, l1)
    assert tr.trainer.lr < 1e-3 and tr.sched_step == 12
    tr._save_checkpoint("epoch_1")
    after = ctrl.diffusion_model.net.state_dict()
    assert any(not torch.equal(before[k].cpu(), after[k].cpu()) for k in before)
    assert ctrl.diffusion_model.ema.num_updates == 12
    ctrl2 = synth.build_controller(DiffusionController, "fp32", device=dev)
    tr2 = DiffusionControllerTrainer(ctrl2, _DataModule(stats), learning_rate=1e-3, checkpoint_dir=str(tmp_path / "ckpt2"), device=dev)
    tr2.load_checkpoint(str(tmp_path / "ckpt" / "epoch_1"))
    for k, v in after.items():
        assert torch.equal(ctrl2.diffusion_model.net.state_dict()[k].cpu(), v.cpu()), k
    for a, b in zip(ctrl.diffusion_model.ema.shadow_params, ctrl2.diffusion_model.ema.shadow_params):
        assert torch.equal(a.cpu(), b.cpu())
    # the checkpoint carries torch_ema's num_updates but (like the reference, which saves no optimizer.pt) no AdamW state: the EMA warm-up
    # counter resumes, the AdamW step t starts again at 0 with its fresh moments
    assert tr2.trainer.ema_updates == 12 and tr2.trainer.step_count == 0
    l2, _ = tr2.eval_step(batch, t, z)
    assert abs(l2 - l1) < 1e-5 * max(1.0, abs(l1)), (l1, l2)
    out = ctrl2.predict(batch["states"][:, 1], batch["vla_actions"], batch["images_cam1"][:, -1], batch["images_cam2"][:, -1], batch["forces"][:, 1])
    assert out.shape == (8, 16, 10) and bool(torch.isfinite(out).all())


# ------------------------------------------------------------------------------------------------ LSTM residual head
def test_lstm_training_step_matches_reference_autograd_adamw():
    """g14: TactileLSTMController.get_loss + backward + AdamW + cosine LR of the REFERENCE (eval-mode dropout), two steps."""
    from vlatouch.train import LstmTrainer
    from tools.make_golden_train_lstm import lstm_train_inputs
    g = np.load(f"{cases.GOLDEN}/g14_train_lstm.npz")
    names = [str(n) for n in g["names"]]
    tr = LstmTrainer(cases.lstm_mods(), lr=1e-4, weight_decay=1e-6, device="cuda:0")
    for step in (1, 2):
        inp = lstm_train_inputs(step)
        tr.lr = 1e-4 if step == 1 else 9.999999997779339e-05
        loss, pred = tr.get_loss(inp["obs_in"], inp["vla_n"], inp["forces"], inp["expert_n"], masks=None)
        assert abs(loss - float(g[f"s{step}_loss"][0])) < 1e-5 * abs(float(g[f"s{step}_loss"][0])), (loss, g[f"s{step}_loss"])
        assert float(np.abs(pred.cpu().numpy() - g[f"s{step}_pred"]).max()) < 2e-5
        e = float(np.abs(tr.last_dcond.cpu().numpy() - g[f"s{step}_dcond"]).max())
        assert e < 1e-4 * float(np.abs(g[f"s{step}_dcond"]).max()), e
        flat = lambda d: {f"{m}.{k}": v for m, sd in d.items() for k, v in sd.items()}
        grads = flat(tr.modules_grads())
        assert set(names) == set(grads), set(names) ^ set(grads)
        wg = check(g[f"s{step}_grad"], names, grads, "grad", 1e-4)
        tr.optimizer_step()
        wp = check(g[f"s{step}_param"], names, flat(tr.modules_state_dict()), "param", 1e-6)
        print(f"[lstm train step {step}] loss {loss:.6f}; worst relative error: grads {wg:.2e}, params {wp:.2e}")


def test_lstm_training_dropout_masks_against_torch_restatement():
    """Training-mode arithmetic with injected keep-masks (inter-layer LSTM dropout, head dropout) against torch autograd on the CPU:
    two single-layer nn.LSTMs with the mask between them are nn.LSTM(num_layers=2, dropout=p) for that mask."""
    from vlatouch.train import LstmTrainer
    from tools.make_golden_train_lstm import lstm_train_inputs
    mods = cases.lstm_mods()
    inp = lstm_train_inputs(3)
    B, T, H, p = 16, 16, 256, 0.1
    gen = torch.Generator().manual_seed(9)
    masks = {"lstm": (torch.rand(B, T, H, generator=gen) >= p).float() / (1 - p), "head": (torch.rand(B, T, H, generator=gen) >= p).float() / (1 - p)}
    with torch.enable_grad():
        nn = torch.nn
        mlp = lambda sd, n: nn.Sequential(*[m for i in range(n) for m in ([nn.Linear(sd[f"{2 * i}.weight"].shape[1], sd[f"{2 * i}.weight"].shape[0])] +
                                                                           ([nn.GELU()] if i + 1 < n else []))])
        obs_e, force_e = mlp(mods["obs_encoder"], 3), mlp(mods["force_encoder"], 2)
        obs_e.load_state_dict(mods["obs_encoder"]), force_e.load_state_dict(mods["force_encoder"])
        l0, l1 = nn.LSTM(138, H, batch_first=True), nn.LSTM(H, H, batch_first=True)
        l0.load_state_dict({k[:-1] + "0": v for k, v in mods["lstm"].items() if k.endswith("l0")})
        l1.load_state_dict({k[:-1] + "0": v for k, v in mods["lstm"].items() if k.endswith("l1")})
        hd = mods["output_head"]
        lin0, ln, lin4 = nn.Linear(2 * H, H), nn.LayerNorm(H), nn.Linear(H, 10)
        lin0.load_state_dict({"weight": hd["0.weight"], "bias": hd["0.bias"]}), ln.load_state_dict({"weight": hd["1.weight"], "bias": hd["1.bias"]})
        lin4.load_state_dict({"weight": hd["4.weight"], "bias": hd["4.bias"]})
        cond = obs_e(inp["obs_in"])
        x = torch.cat([force_e(inp["forces"].reshape(B * T, 3)).reshape(B, T, -1), inp["vla_n"]], -1)
        h0, _ = l0(x)
        h1, _ = l1(h0 * masks["lstm"])
        a = torch.nn.functional.gelu(ln(lin0(torch.cat([h1, cond[:, None].expand(B, T, H)], -1)))) * masks["head"]
        pred = inp["vla_n"] + lin4(a)
        loss = torch.nn.functional.mse_loss(pred, inp["expert_n"])
        loss.backward()
    tr = LstmTrainer(mods, device="cuda:0")
    got, gp = tr.get_loss(inp["obs_in"], inp["vla_n"], inp["forces"], inp["expert_n"], masks=masks)
    assert abs(got - float(loss)) < 1e-5 * float(loss) and float((gp.cpu() - pred.detach()).abs().max()) < 2e-5
    G = tr.modules_grads()
    ref = {"lstm.weight_ih_l0": l0.weight_ih_l0.grad, "lstm.weight_hh_l0": l0.weight_hh_l0.grad, "lstm.bias_hh_l0": l0.bias_hh_l0.grad,
           "lstm.weight_ih_l1": l1.weight_ih_l0.grad, "lstm.weight_hh_l1": l1.weight_hh_l0.grad, "output_head.0.weight": lin0.weight.grad,
           "output_head.1.weight": ln.weight.grad, "output_head.1.bias": ln.bias.grad, "output_head.4.weight": lin4.weight.grad,
           "force_encoder.0.weight": force_e[0].weight.grad, "obs_encoder.0.weight": obs_e[0].weight.grad, "obs_encoder.4.bias": obs_e[4].bias.grad}
    for k, want in ref.items():
        m, kk = k.split(".", 1)
        e = float((G[m][kk] - want).norm() / want.norm())
        assert e < 1e-4, (k, e)
    # 'draw' masks: a step runs, the loss is finite, parameters move
    before = tr.modules_state_dict()["lstm"]["weight_hh_l1"].clone()
    l = tr.train_step(inp["obs_in"], inp["vla_n"], inp["forces"], inp["expert_n"], masks="draw")
    assert np.isfinite(l) and not torch.equal(before, tr.modules_state_dict()["lstm"]["weight_hh_l1"])


def test_lstm_trainer_mirror_overfits_a_batch_and_round_trips(tmp_path):
    """LSTMControllerTrainer (lstm_train.py:19): fixed batch -> the loss falls; trained tensors reach the controller, whose
    inference kernels (predict_sequence / forward) then reproduce the trainer's own prediction; save -> load keeps them."""
    from vlatouch import synth
    from residual_controller.lstm_step_controller import TactileLSTMController
    from residual_controller.lstm_train import LSTMControllerTrainer
    dev = "cuda:0"
    c = synth.DINOV2_CONFIGS["small"]
    mk = lambda: TactileLSTMController(device=dev, precision="fp32",
                                       image_state_dict=synth.torch_state_dict(synth.dinov2_shapes(c["hidden"], c["layers"]), prefix="dinov2-small."))
    ctrl = mk()
    stats = {k: v.cpu().numpy() for k, v in synth.unit_stats().items()}
    tr = LSTMControllerTrainer(ctrl, _DataModule(stats), learning_rate=1e-3, checkpoint_dir=str(tmp_path / "ck"), device=dev)
    batch = _batch(6)
    batch["forces"] = batch["forces"][:, :18]                  # context 2 + horizon 16: forces[:, 1:-1] is the 16-tick window
    l0 = tr.eval_step(batch)
    losses = [tr.train_step(batch, masks=None) for _ in range(10)]
    l1 = tr.eval_step(batch)
    assert np.isfinite(losses).all() and l1 < 0.9 * l0, (l0, losses, l1)
    tr._save_checkpoint("epoch_1")
    bd = tr._prepare_batch(batch)
    _, pred = tr.trainer.get_loss(bd["obs_in"], bd["vla_act"], bd["forces"], bd["expert_act"], masks=None, backward=False)
    obs_cond = ctrl.encode_observation(batch["states"][:, 1], batch["images_cam1"][:, -1], batch["images_cam2"][:, -1])
    fwd = ctrl.forward({"vla_act": bd["vla_act"], "obs_cond": obs_cond, "forces": bd["forces"]})
    assert float((fwd - pred).abs().max()) < 2e-4, float((fwd - pred).abs().max())
    lm = float(ctrl.get_loss({"vla_act": bd["vla_act"], "obs_cond": obs_cond, "forces": bd["forces"], "expert_act": bd["expert_act"]}))
    assert abs(lm - l1) < 1e-4 * max(1.0, l1)
    ctrl2 = mk()
    tr2 = LSTMControllerTrainer(ctrl2, _DataModule(stats), learning_rate=1e-3, checkpoint_dir=str(tmp_path / "ck2"), device=dev)
    tr2.load_checkpoint(str(tmp_path / "ck" / "epoch_1"))
    assert abs(tr2.eval_step(batch) - l1) < 1e-6 * max(1.0, l1)


def test_captured_training_step_equals_eager():
    """SITrainer.capture / replay (one hipGraph per step, step-dependent scalars in device memory) == the eager step, bit for bit."""
    from vlatouch.train import SITrainer
    mk = lambda: SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), lr=1e-3, device="cuda:0")
    a, b = mk(), mk()
    b.capture(16)
    for step in (1, 2, 3):
        inp = train_inputs(step)
        args = [inp[k].to("cuda:0") for k in ("obs_in", "vla_n", "expert_n", "t", "z")]
        la, info = a.train_step(*args)
        lb = b.replay(*args)
        torch.cuda.synchronize()
        assert all(float(lb[k]) == info[k] for k in info), (step, info, {k: float(v) for k, v in lb.items()})
    pa, pb = a.net_state_dict(), b.net_state_dict()
    assert all(torch.equal(pa[k], pb[k]) for k in pa)
    ea, eb = a.ema_state_dict(), b.ema_state_dict()
    assert all(torch.equal(ea[k], eb[k]) for k in ea)
    assert all(torch.equal(x, y) for x, y in zip(a.mlp.state_dict().values(), b.mlp.state_dict().values()))


def test_captured_lstm_training_step_equals_eager():
    from vlatouch.train import LstmTrainer
    from tools.make_golden_train_lstm import lstm_train_inputs
    a, b = LstmTrainer(cases.lstm_mods(), lr=1e-3, device="cuda:0"), LstmTrainer(cases.lstm_mods(), lr=1e-3, device="cuda:0")
    b.capture(16, masks=None)
    for step in (1, 2, 3):
        inp = lstm_train_inputs(step)
        args = [inp[k].to("cuda:0") for k in ("obs_in", "vla_n", "forces", "expert_n")]
        la = a.train_step(*args, masks=None)
        lb = b.replay(*args)
        torch.cuda.synchronize()
        assert float(lb) == la, (step, la, float(lb))
    sa, sb = a.modules_state_dict(), b.modules_state_dict()
    assert all(torch.equal(sa[m][k], sb[m][k]) for m in sa for k in sa[m])
    b2 = LstmTrainer(cases.lstm_mods(), device="cuda:0")
    b2.capture(16)                                           # masks drawn inside the graph: a fresh draw per replay
    args = [lstm_train_inputs(1)[k].to("cuda:0") for k in ("obs_in", "vla_n", "forces", "expert_n")]
    l1 = float(b2.replay(*args, schedule=False)); p1 = b2._graph_pred.clone()
    b3 = LstmTrainer(cases.lstm_mods(), device="cuda:0")
    b3.capture(16)
    l2 = float(b3.replay(*args, schedule=False))
    torch.cuda.synchronize()
    assert np.isfinite([l1, l2]).all() and l1 != l2 and not torch.equal(p1, b3._graph_pred)      # same weights and inputs, different masks


def test_trainer_runs_an_epoch_from_h5_episodes(tmp_path):
    """The reference's training entry point end to end on its own on-disk format: ControllerDataModule over episode_*.h5 ->
    DiffusionControllerTrainer.train (one epoch: shuffled batches, device step, validation, best_model / epoch_1 checkpoints)."""
    import os, shutil
    from vlatouch import synth
    from residual_controller.bridge_controller import DiffusionController
    from residual_controller.bridge_train import DiffusionControllerTrainer
    from residual_controller.controller_dataset import ControllerDataModule
    d = tmp_path / "data"
    d.mkdir()
    for f in ("episode_1.h5", "episode_2.h5"):
        shutil.copy(os.path.join(cases.GOLDEN, "episodes_h5", f), d)
    np.random.seed(1)
    torch.manual_seed(1)
    dm = ControllerDataModule(str(d), batch_size=8, num_workers=0, context_frames=2, horizon=16, use_images=True)
    ctrl = synth.build_controller(DiffusionController, "fp32", device="cuda:0")
    tr = DiffusionControllerTrainer(ctrl, dm, learning_rate=1e-4, checkpoint_dir=str(tmp_path / "ck"), device="cuda:0")
    best = tr.train(dm, num_epochs=1, save_interval=1, eval_interval=1, log_interval=1)
    assert np.isfinite(best)
    steps = [h for h in tr.history if "step" in h]
    assert len(steps) == len(dm.train_dataset) // 8 and all(np.isfinite(h["loss"]) for h in steps)
    for name in ("best_model", "epoch_1"):
        assert sorted(os.listdir(tmp_path / "ck" / name)) == ["bridge_model.pt", "controller.pt"]
    ck = torch.load(tmp_path / "ck" / "epoch_1" / "controller.pt", weights_only=False)
    assert set(ck["stats"].keys()) >= {"action_mins", "vla_maxs"}


def test_ragged_batch_is_padded_with_zero_weight_samples():
    """B = 3 (a validation loader's last batch): same losses and gradients as the 3 samples repeated 4 times (B = 12, aligned)."""
    from vlatouch.train import SITrainer
    inp = train_inputs(1)
    a = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), device="cuda:0")
    b = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), device="cuda:0")
    small = [inp[k][:3] for k in ("obs_in", "vla_n", "expert_n", "t", "z")]
    rep = [torch.cat([x] * 4) for x in small]
    la, ia = a.get_loss(*small)
    lb, ib = b.get_loss(*rep)
    assert abs(la - lb) < 2e-6 * abs(lb), (la, lb)
    assert a.last_dcond.shape == (3, 256)
    ga, gb = dict(a.net_grads()), dict(b.net_grads())
    worst = max(float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-20)) for k in ga)
    assert worst < 2e-5, worst


def test_lstm_ragged_batch_is_padded_with_zero_weight_samples():
    from vlatouch.train import LstmTrainer
    from tools.make_golden_train_lstm import lstm_train_inputs
    inp = lstm_train_inputs(1)
    a, b = LstmTrainer(cases.lstm_mods(), device="cuda:0"), LstmTrainer(cases.lstm_mods(), device="cuda:0")
    small = [inp[k][:3] for k in ("obs_in", "vla_n", "forces", "expert_n")]
    la, pa = a.get_loss(*small)
    lb, pb = b.get_loss(*[torch.cat([x] * 4) for x in small])
    assert abs(la - lb) < 2e-6 * abs(lb) and pa.shape == (3, 16, 10) and float((pa - pb[:3]).abs().max()) < 1e-6
    flat = lambda d: {f"{m}.{k}": v for m, sd in d.items() for k, v in sd.items()}
    ga, gb = flat(a.modules_grads()), flat(b.modules_grads())
    worst = max(float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-20)) for k in ga)
    assert worst < 2e-5, worst


def test_training_step_at_bench_batch_is_the_mean_of_its_sub_batches():
    """Full-size property (B = 128, the reference's default batch and tools/train_bench.py's): every loss is a batch mean, so the
    losses and gradients of the 128-sample step equal the average over its eight 16-sample sub-batches (whose arithmetic g13 pins)."""
    from vlatouch.train import SITrainer
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(11)
    B = 128
    r = lambda *s: torch.randn(*s, generator=gen)
    obs, x0, x1, t, z = r(B, 781), r(B, 16, 10).clamp(-1, 1), r(B, 16, 10).clamp(-1, 1), torch.rand(B, generator=gen), r(B, 16, 10)
    big = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), device=dev)
    lb, ib = big.get_loss(obs, x0, x1, t, z)
    gb = dict(big.net_grads())
    gb.update({"state_encoder." + k: v for k, v in big.mlp.grads().items()})
    small = SITrainer(cases.si_net_sd(""), cases.state_encoder_sd(781), device=dev)
    acc, lsum = None, 0.0
    for i in range(0, B, 16):
        sl = slice(i, i + 16)
        l, _ = small.get_loss(obs[sl], x0[sl], x1[sl], t[sl], z[sl])
        lsum += l
        g = dict(small.net_grads())
        g.update({"state_encoder." + k: v for k, v in small.mlp.grads().items()})
        acc = {k: v.double() for k, v in g.items()} if acc is None else {k: acc[k] + g[k].double() for k in g}
    assert abs(lb - lsum / 8) < 2e-6 * abs(lb), (lb, lsum / 8)
    worst = max(float((gb[k].double() - acc[k] / 8).norm() / (acc[k].norm() / 8 + 1e-30)) for k in gb)
    assert worst < 5e-5, worst
