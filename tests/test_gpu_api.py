"""GPU parity at the drop-in boundary: the mirror classes of vla-touch_amd/residual_controller (same names and
signatures as the reference) against golden vectors captured from the reference, plus checkpoint round trips and
the reference's error behaviour."""
import os
import tempfile

import numpy as np
import pytest
import torch

from tests import cases
from tests.test_oracle_golden import SCHEDULE_CASES

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = {"fp32": 1e-4, "bf16": 1e-2}


def G(name):
    return np.load(f"{cases.GOLDEN}/{name}.npz")


def err(a, b):
    return float(np.abs(a.detach().float().cpu().numpy().astype(np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.fixture(scope="module")
def controllers():
    from residual_controller.bridge_controller import DiffusionController
    return {p: cases.build_controller(DiffusionController, precision=p) for p in ("fp32", "bf16")}


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_predict_end_to_end_golden(controllers, prec):
    g = G("g5_predict_e2e")
    ctrl = controllers[prec]
    inp = cases.predict_inputs(2, 16, 224)
    obs = ctrl.encode_observation(inp["state"], inp["cam1"], inp["cam2"], inp["forces"])
    pred = ctrl.predict(inp["state"], inp["vla"], inp["cam1"], inp["cam2"], inp["forces"], noise=torch.from_numpy(g["z"]))
    assert pred.shape == (2, 16, 10) and pred.dtype == torch.float32
    e_obs, e_pred = err(obs, g["obs_cond"]), err(pred, g["pred"])
    print(f"[{prec}] obs_cond err {e_obs:.3e}  a_hat err {e_pred:.3e}")
    assert e_obs < (2e-4 if prec == "fp32" else 5e-3), e_obs
    # north_star tolerance on a_hat: 1e-4 fp32, 1e-2 low precision ("bf16" mode = fp16 DINOv2 + bf16 MLP + split-bf16 U-Nets)
    assert e_pred < TOL[prec], e_pred


def test_predict_draws_its_own_noise(controllers):
    ctrl = controllers["fp32"]
    inp = cases.predict_inputs(2, 16, 224)
    torch.manual_seed(0)
    a = ctrl.predict(inp["state"], inp["vla"], inp["cam1"], inp["cam2"], inp["forces"])
    torch.manual_seed(0)
    b = ctrl.predict(inp["state"], inp["vla"], inp["cam1"], inp["cam2"], inp["forces"])
    torch.manual_seed(1)
    c = ctrl.predict(inp["state"], inp["vla"], inp["cam1"], inp["cam2"], inp["forces"])
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_checkpoint_roundtrip_reference_format(controllers):
    from residual_controller.bridge_controller import load_bridge_controller
    ctrl = controllers["fp32"]
    inp = cases.predict_inputs(2, 16, 224)
    z = torch.from_numpy(G("g5_predict_e2e")["z"])
    ref = ctrl.predict(inp["state"], inp["vla"], inp["cam1"], inp["cam2"], inp["forces"], noise=z)
    with tempfile.TemporaryDirectory() as d:
        saved_stats = ctrl.stats
        ctrl.stats = {k: v.cpu().numpy() for k, v in saved_stats.items()}      # the reference stores numpy stats
        ctrl.save(d)
        ctrl.stats = saved_stats
        ck = torch.load(os.path.join(d, "controller.pt"), weights_only=False)
        assert sorted(ck.keys()) == ["force_decoder", "model_args", "state_encoder", "stats"]
        bm = torch.load(os.path.join(d, "bridge_model.pt"), weights_only=False)
        assert sorted(bm.keys()) == ["ema", "net"] and sorted(bm["ema"].keys()) == ["collected_params", "decay", "num_updates", "shadow_params"]
        assert len(bm["ema"]["shadow_params"]) == 438
        new = load_bridge_controller(device="cuda:0", precision="fp32", image_state_dict=cases.dino_sd("small"))
        new.load(d)
        assert all(torch.is_tensor(v) and v.is_cuda and v.dtype == torch.float32 for v in new.stats.values())
        out = new.predict(inp["state"], inp["vla"], inp["cam1"], inp["cam2"], inp["forces"], noise=z)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("tag,sde,gt,et", SCHEDULE_CASES)
def test_other_schedules_and_bs_integrator(controllers, tag, sde, gt, et):
    g = G(f"g2_si_{tag}")
    si = controllers["fp32"].diffusion_model
    old = (si.sde_type, si.gamma_type, si.epsilon_type)
    try:
        si.sde_type, si.gamma_type, si.epsilon_type = sde, gt, et
        x0, cond, _ = cases.si_inputs(2, 16)
        xT, traj = si.sample(x0, cond, diffuse_step=8, recod_traj=True, noise=torch.from_numpy(g["z"]))
        assert len(traj) == 9
        assert err(torch.stack(traj), g["traj"]) < 1e-4, (tag, err(torch.stack(traj), g["traj"]))
    finally:
        si.sde_type, si.gamma_type, si.epsilon_type = old


@pytest.mark.parametrize("tag,sde,sw,direction", [("vs_backward_w07", "vs", 0.7, "backward"), ("bs_backward_w13", "bs", 1.3, "backward"),
                                                  ("vs_forward_w05", "vs", 0.5, "forward")])
def test_backward_direction_and_score_weight(controllers, tag, sde, sw, direction):
    """sde_vs / sde_bs of the mirror with direction='backward' and score_weight != 1 against the reference's own run (g2 variants)."""
    g = G(f"g2_si_{tag}")
    si = controllers["fp32"].diffusion_model
    x0, cond, _ = cases.si_inputs(2, 16)
    fn = si.sde_bs if sde == "bs" else si.sde_vs
    xT, traj = fn(x_initial=x0, cond=cond, delta_t=float(1.0 / 8), score_weight=sw, direction=direction, noise=torch.from_numpy(g["z"]))
    assert len(traj) == 9
    scale = max(1.0, float(np.abs(g["traj"]).max()))       # the backward SDE blows up in its last step (|x| ~ 77): relative tolerance there
    assert err(torch.stack(traj[:-1]), g["traj"][:-1]) < 1e-4, (tag, err(torch.stack(traj[:-1]), g["traj"][:-1]))
    assert err(torch.stack(traj), g["traj"]) < 1e-4 * scale, (tag, err(torch.stack(traj), g["traj"]))
    assert err(xT, g["traj"][-1]) < 1e-4 * scale
    with pytest.raises(NotImplementedError):
        si.sde_vs(x_initial=x0, cond=cond, delta_t=0.125, direction="sideways")


def test_error_behaviour_matches_reference(controllers):
    from residual_controller.bridge.bridge_model import StochasticInterpolants
    from residual_controller.controller_dataset import normalize_actions
    si = controllers["fp32"].diffusion_model
    x0, cond, _ = cases.si_inputs(2, 16)
    old = si.gamma_type
    si.gamma_type = "nope"
    with pytest.raises(NotImplementedError):
        si.gamma(torch.tensor(0.5))
    with pytest.raises(NotImplementedError):
        si.sample(x0, cond)
    si.gamma_type = old
    old = si.sde_type
    si.sde_type = "xx"
    with pytest.raises(NotImplementedError):
        si.sample(x0, cond)
    si.sde_type = old
    with pytest.raises(ValueError):
        normalize_actions(x0, cases.stats(), "bogus")
    with pytest.raises(NotImplementedError):
        StochasticInterpolants().load_model({**cases.MODEL_ARGS, "net_type": "other"}, "cuda:0")
    with pytest.raises(RuntimeError):
        controllers["fp32"].state_encoder.load_state_dict({"0.weight": torch.zeros(3, 3)})


def test_unet_module_call_contract(controllers):
    """net.v_net(sample, timestep, global_cond=cond) as the reference's sde loop calls it (bridge_model.py:352-353)."""
    g = G("g1_unet_fwd")
    net = controllers["fp32"].diffusion_model.net
    x, cond = cases.unet_inputs(2, 16)
    v = net.v_net(x, torch.full((2,), 0.5), global_cond=cond)
    s = net.s_net(x.cuda(), 0.5, global_cond=cond)
    assert err(v, g["v_B2_T16_t0.5"]) < 1e-4 and err(s, g["s_B2_T16_t0.5"]) < 1e-4


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_lstm_controller_golden(prec):
    from residual_controller.lstm_step_controller import TactileLSTMController
    g = G("g6_lstm")
    c = TactileLSTMController(device="cuda:0", precision=prec, image_state_dict=cases.dino_sd("small"))
    for name, sd in cases.lstm_mods(384).items():
        getattr(c, name).load_state_dict(sd)
    c.to("cuda:0")
    c.stats = cases.stats("nontrivial")
    li = cases.lstm_inputs(2, 16)
    seq = c.predict_sequence(li["obs_cond"], li["vla"], li["forces"])
    assert err(seq, g["predict_sequence"]) < 1e-4
    assert err(c.hidden_state, g["h"]) < 1e-4 and err(c.cell_state, g["c"]) < 1e-4
    from residual_controller.controller_dataset import normalize_actions
    vn = normalize_actions(li["vla"], c.stats, "vla")
    assert err(c.forward({"vla_act": vn, "obs_cond": li["obs_cond"], "forces": li["forces"]}), g["forward"]) < 1e-4
    pi = cases.predict_inputs(2, 16, 224)
    obs = c.encode_observation(pi["state"], pi["cam1"], pi["cam2"])
    assert err(obs, g["obs_cond"]) < (2e-4 if prec == "fp32" else 3e-2)
    with tempfile.TemporaryDirectory() as d:
        c.save(d)
        c2 = TactileLSTMController(device="cuda:0", precision=prec, image_state_dict=cases.dino_sd("small"))
        c2.load(d)
        assert torch.equal(c2.predict_sequence(li["obs_cond"], li["vla"], li["forces"]), seq)


@pytest.mark.parametrize("hidden,layers", [(128, 2), (384, 3)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_lstm_controller_other_widths_golden(prec, hidden, layers):
    """`lstm_train.py --hidden_dim` 128 / 384 (2 / 3 LSTM layers): the persistent sequence kernel templated on the hidden size, against the reference
    class's own run (g6_lstm_h*: forward, predict_sequence with carried state, observation encoding); ragged batch (3 rows of a 16-row block)."""
    from residual_controller.lstm_step_controller import TactileLSTMController
    from residual_controller.controller_dataset import normalize_actions
    g = G(f"g6_lstm_h{hidden}l{layers}")
    c = TactileLSTMController(hidden_dim=hidden, num_layers=layers, device="cuda:0", precision=prec, image_state_dict=cases.dino_sd("small"))
    for name, sd in cases.lstm_mods(384, hidden=hidden, layers=layers).items():
        getattr(c, name).load_state_dict(sd)
    c.to("cuda:0")
    c.stats = cases.stats("nontrivial")
    li = cases.lstm_inputs(3, 16, hidden=hidden)
    seq = c.predict_sequence(li["obs_cond"], li["vla"], li["forces"])
    assert err(seq, g["predict_sequence"]) < 1e-4, err(seq, g["predict_sequence"])
    assert err(c.hidden_state, g["h"]) < 1e-4 and err(c.cell_state, g["c"]) < 1e-4
    vn = normalize_actions(li["vla"], c.stats, "vla")
    assert err(c.forward({"vla_act": vn, "obs_cond": li["obs_cond"], "forces": li["forces"]}), g["forward"]) < 1e-4
    pi = cases.predict_inputs(2, 16, 224)
    assert err(c.encode_observation(pi["state"], pi["cam1"], pi["cam2"]), g["obs_cond"]) < (2e-4 if prec == "fp32" else 3e-2)


def test_lstm_controller_rejects_widths_the_kernel_cannot_deal():
    from residual_controller.lstm_step_controller import TactileLSTMController
    for bad in (64, 200, 512):
        with pytest.raises(ValueError):
            TactileLSTMController(hidden_dim=bad, device="cuda:0", image_state_dict=cases.dino_sd("small"))


def test_no_cpu_fallback():
    from vlatouch import _lib
    with pytest.raises(_lib.VtError):
        _lib.require_gpu("cpu")


@pytest.mark.parametrize("B,T,res,layout", [(1, 32, 384, "bthwc"), (3, 8, 224, "uint8_bhwc"), (5, 4, 224, "bchw"), (2, 16, 518, "bchw")])
def test_predict_edge_shapes_vs_oracle(controllers, B, T, res, layout):
    """Shapes off the golden set, against the oracle run live: BASELINE configs[0] (batch 1, T = 32, 384x384 frames in the
    reference's 5-D [B,1,H,W,3] layout), odd batches, the shortest horizon the U-Net accepts (T = 4), uint8 BHWC frames (the
    /255 branch) and the native 518 resolution (no position-embedding interpolation)."""
    from oracle import controller as oc
    ctrl = controllers["fp32"]
    g = np.random.default_rng(100 + B)
    state = torch.from_numpy(g.standard_normal((B, 10)).astype(np.float32))
    forces = torch.from_numpy(g.standard_normal((B, 3)).astype(np.float32))
    vla = torch.from_numpy(g.uniform(0, 1, (B, T, 10)).astype(np.float32))
    z = torch.from_numpy(g.standard_normal((10, B, T, 10)).astype(np.float32))
    u1, u2 = 0.2 + 0.8 * g.random((B, 3, res, res)), 0.6 * g.random((B, 3, res, res))
    if layout == "bchw":
        cam1, cam2 = torch.from_numpy(u1.astype(np.float32)), torch.from_numpy(u2.astype(np.float32))
    elif layout == "uint8_bhwc":
        cam1 = torch.from_numpy((255 * u1).astype(np.uint8).transpose(0, 2, 3, 1).copy())
        cam2 = torch.from_numpy((255 * u2).astype(np.uint8).transpose(0, 2, 3, 1).copy())
    else:
        cam1 = torch.from_numpy(u1.astype(np.float32).transpose(0, 2, 3, 1)[:, None].copy())
        cam2 = torch.from_numpy(u2.astype(np.float32).transpose(0, 2, 3, 1)[:, None].copy())
    ref = oc.predict(cases.dino_sd("small"), 6, cases.state_encoder_sd(781), cases.si_net_sd("ema"), cases.stats("nontrivial"),
                     state, vla, cam1, cam2, forces, z)
    got = ctrl.predict(state.cuda(), vla.cuda(), cam1.cuda(), cam2.cuda(), forces.cuda(), noise=z.cuda())
    assert got.shape == (B, T, 10)
    assert err(got, ref.numpy()) < TOL["fp32"], err(got, ref.numpy())


@pytest.mark.parametrize("B,size,force_dim", [(2, "small", 64), (32, "base", 64), (4, "small", 17)])
def test_predict_tactile_vector_widths_vs_oracle(B, size, force_dim):
    """The tactile input m_t at other widths than the marker tracker's 3-d force estimate: `force_dim` is a constructor argument of the
    reference (bridge_controller.py:25; obs_dim = 2*latent + state_dim + force_dim, :40-48; cat at :129-132) and BASELINE.json's synthetic
    workload names a 64-d tactile vector.  B = 32 / DINOv2-base / 64-d is that workload's pi_I leg exactly; 17 makes the concatenated
    row end off every padding boundary of the observation MLP's first Linear.  Both precisions against the oracle run live."""
    from oracle import controller as oc
    from residual_controller.bridge_controller import DiffusionController
    T = 16
    g = np.random.default_rng(6400 + B + force_dim)
    state = torch.from_numpy(g.standard_normal((B, 10)).astype(np.float32))
    forces = torch.from_numpy(g.standard_normal((B, force_dim)).astype(np.float32))
    vla = torch.from_numpy(g.uniform(0, 1, (B, T, 10)).astype(np.float32))
    z = torch.from_numpy(g.standard_normal((10, B, T, 10)).astype(np.float32))
    cam1 = torch.from_numpy((0.2 + 0.8 * g.random((B, 3, 224, 224))).astype(np.float32))
    cam2 = torch.from_numpy((0.6 * g.random((B, 3, 224, 224))).astype(np.float32))
    latent, heads = (384, 6) if size == "small" else (768, 12)
    obs_dim = 2 * latent + 10 + force_dim
    if B > 8:                                                                  # DINOv2-base on 64 frames: give the oracle the host's cores
        torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    ref = oc.predict(cases.dino_sd(size), heads, cases.state_encoder_sd(obs_dim), cases.si_net_sd("ema"), cases.stats("nontrivial"),
                     state, vla, cam1, cam2, forces, z)
    for prec in ("fp32", "bf16"):
        ctrl = cases.build_controller(DiffusionController, precision=prec, size=size, force_dim=force_dim)
        assert ctrl.obs_dim == obs_dim and ctrl.state_encoder.state_dict()["0.weight"].shape == (256, obs_dim)
        got = ctrl.predict(state.cuda(), vla.cuda(), cam1.cuda(), cam2.cuda(), forces.cuda(), noise=z.cuda())
        e = err(got, ref.numpy())
        print(f"[force_dim {force_dim} B {B} dinov2-{size} {prec}] a_hat err {e:.3e}")
        assert got.shape == (B, T, 10) and e < TOL[prec], (prec, e)
        # the tactile vector is really consumed: another m_t gives another a_hat
        got2 = ctrl.predict(state.cuda(), vla.cuda(), cam1.cuda(), cam2.cuda(), (forces + 1.0).cuda(), noise=z.cuda())
        assert float((got2 - got).abs().max()) > 1e-4
        del ctrl


@pytest.mark.parametrize("B", [1, 6])
def test_predict_48_tick_chunks_vs_oracle(controllers, B):
    """The reference's second robot cadence: scripts/franka_inference_eef.py refines 48-tick chunks.  End to end against the oracle in
    both precisions; in the low-precision mode the sampler's U-Nets run on the FUSED path at T = 48 (levels 48 / 24 / 12 on 48-row blocks)."""
    from oracle import controller as oc
    from vlatouch import _lib as L
    T = 48
    g = np.random.default_rng(480 + B)
    state = torch.from_numpy(g.standard_normal((B, 10)).astype(np.float32))
    forces = torch.from_numpy(g.standard_normal((B, 3)).astype(np.float32))
    vla = torch.from_numpy(g.uniform(0, 1, (B, T, 10)).astype(np.float32))
    z = torch.from_numpy(g.standard_normal((10, B, T, 10)).astype(np.float32))
    cam1 = torch.from_numpy((0.2 + 0.8 * g.random((B, 3, 224, 224))).astype(np.float32))
    cam2 = torch.from_numpy((0.6 * g.random((B, 3, 224, 224))).astype(np.float32))
    ref = oc.predict(cases.dino_sd("small"), 6, cases.state_encoder_sd(781), cases.si_net_sd("ema"), cases.stats("nontrivial"),
                     state, vla, cam1, cam2, forces, z)
    for prec in ("fp32", "bf16"):
        ctrl = controllers[prec]
        got = ctrl.predict(state.cuda(), vla.cuda(), cam1.cuda(), cam2.cuda(), forces.cuda(), noise=z.cuda())
        assert got.shape == (B, T, 10)
        assert err(got, ref.numpy()) < TOL[prec], (prec, err(got, ref.numpy()))
    eng = controllers["bf16"].diffusion_model._sampler[1]                # the U-Net pair engine the predict() above sampled with
    assert L.lib().vt_unet_fused_covers(eng._h, B, T, 10) == 1


def test_predict_rejects_horizons_the_unet_cannot_run(controllers):
    """T must be divisible by 4 (two stride-2 downsamplings, conditional_unet_1D.py): the reference fails inside torch.cat on
    the skip connection; here the driver refuses up front."""
    ctrl = controllers["fp32"]
    inp = cases.predict_inputs(2, 16, 224)
    with pytest.raises(Exception):
        ctrl.predict(inp["state"].cuda(), inp["vla"][:, :6].cuda(), inp["cam1"].cuda(), inp["cam2"].cuda(), inp["forces"].cuda())


def test_predict_and_rdt_chunk_are_graph_capturable(controllers):
    """The C ABI promises: no allocation, no synchronisation, everything on the caller's stream.  So a whole predict() (and an RDT
    predict_action) must be capturable in a hipGraph after one warm-up call sized the workspaces, and replays must reproduce the
    eager result bit for bit (how bench.py runs the step)."""
    from models.rdt_runner import RDTRunner
    ctrl = controllers["bf16"]
    inp = {k: v.cuda() for k, v in cases.predict_inputs(4, 16, 224).items()}
    z = inp["z"].repeat(1, 2, 1, 1)[:, :4].contiguous()
    eager = ctrl.predict(inp["state"][:4] if inp["state"].shape[0] >= 4 else inp["state"].repeat(2, 1), inp["vla"].repeat(2, 1, 1)[:4],
                         inp["cam1"].repeat(2, 1, 1, 1)[:4], inp["cam2"].repeat(2, 1, 1, 1)[:4], inp["forces"].repeat(2, 1)[:4], noise=z)
    args = (inp["state"].repeat(2, 1)[:4].contiguous(), inp["vla"].repeat(2, 1, 1)[:4].contiguous(), inp["cam1"].repeat(2, 1, 1, 1)[:4].contiguous(),
            inp["cam2"].repeat(2, 1, 1, 1)[:4].contiguous(), inp["forces"].repeat(2, 1)[:4].contiguous())
    stream = torch.cuda.Stream()
    holder = {}
    with torch.cuda.stream(stream):
        holder["out"] = ctrl.predict(*args, noise=z)           # warm-up on the capture stream
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            holder["out"] = ctrl.predict(*args, noise=z)
        for _ in range(2):
            g.replay()
        stream.synchronize()
    assert torch.equal(holder["out"], eager)
    # RDT chunk (tiny config), noise injected
    cfg = cases.RDT_TINY
    c = {"rdt": {"hidden_size": cfg["hidden"], "depth": cfg["depth"], "num_heads": cfg["heads"]}, "lang_adaptor": "mlp2x_gelu",
         "img_adaptor": "mlp2x_gelu", "state_adaptor": "mlp3x_gelu",
         "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": 5, "beta_schedule": "squaredcos_cap_v2",
                             "prediction_type": "sample", "clip_sample": False}}
    r = RDTRunner(action_dim=cfg["action_dim"], pred_horizon=cfg["horizon"], config=c, lang_token_dim=cfg["lang_token_dim"],
                  img_token_dim=cfg["img_token_dim"], state_token_dim=cfg["state_token_dim"], max_lang_cond_len=cfg["max_lang_cond_len"],
                  img_cond_len=cfg["img_cond_len"], dtype=torch.bfloat16, device="cuda:0")
    r.load_state_dict(cases.rdt_sd(cfg, torch.float32))
    ri = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cases.rdt_inputs(cfg, 2, 12, dtype=torch.bfloat16).items()}
    call = lambda: r.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"].cuda(),
                                    x_init=ri["x_init"])
    eager = call()
    with torch.cuda.stream(stream):
        holder["rdt"] = call()
        stream.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=stream):
            holder["rdt"] = call()
        g2.replay()
        stream.synchronize()
    assert torch.equal(holder["rdt"], eager)


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("x3", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("B,T", [(1, 1), (19, 5), (33, 16)])
def test_lstm_persistent_kernel_ragged_batches_vs_oracle(prec, tol, B, T):
    """The one-launch LSTM head (csrc/vt_lstm.hip) on batch sizes that are not multiples of its 16-row blocks (1, 19, 33 rows), a single
    tick and a 16-tick chunk, in its three arithmetic modes, against the oracle's step loop; T ticks in one launch == T one-tick
    launches with the carried state, bit for bit (lstm_step_controller.py:232-319)."""
    from oracle import controller as oc
    from vlatouch.engine import LstmEngine
    mods = cases.lstm_mods(384)
    eng = LstmEngine({k: mods[k] for k in ("force_encoder", "lstm", "output_head")}, precision=prec, device="cuda:0")
    g = synth_rng = __import__("vlatouch.synth", fromlist=["x"]).inputs_rng(40 + B)
    obs = cases.T(g.standard_normal((B, 256), dtype=np.float32))
    vla = cases.T(g.uniform(-1, 1, (B, T, 10)).astype(np.float32))
    force = cases.T(g.standard_normal((B, T, 3), dtype=np.float32))
    h0 = cases.T(0.1 * g.standard_normal((2, B, 256), dtype=np.float32))
    c0 = cases.T(0.1 * g.standard_normal((2, B, 256), dtype=np.float32))
    h, c = h0.cuda().clone(), c0.cuda().clone()
    out = eng.sequence(obs, vla, force, h, c)
    rh, rc, ref = h0.clone(), c0.clone(), []
    for t in range(T):
        o, rh, rc = oc.lstm_step(mods, obs, vla[:, t], force[:, t], rh, rc, 2)
        ref.append(o)
    ref = torch.stack(ref, dim=1)
    assert out.shape == (B, T, 10)
    assert err(out, ref.numpy()) < tol and err(h, rh.numpy()) < tol and err(c, rc.numpy()) < tol, (err(out, ref.numpy()), err(h, rh.numpy()))
    h2, c2 = h0.cuda().clone(), c0.cuda().clone()
    steps = torch.stack([eng.step(obs, vla[:, t], force[:, t], h2, c2) for t in range(T)], dim=1)
    assert torch.equal(steps, out) and torch.equal(h2, h) and torch.equal(c2, c)
    with pytest.raises(ValueError):
        eng.sequence(obs[:-1] if B > 1 else obs.repeat(2, 1), vla, force, h, c)
