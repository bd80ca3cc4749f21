"""Full-size (BASELINE.json configs[3] shapes) checks through size-independent properties — the oracle cannot run RDT-1B
in test time, so these assert what must hold at any size:
  * batch invariance: a sample's result does not depend on which other samples share the batch (independent episodes are
    what the multi-GPU sharding relies on), including the first/last-tile masking of the condition K/V tile stream;
  * determinism: the same inputs + noise give bit-identical outputs on repeated calls;
  * masked language tokens have no influence.
Sizes: RDT-1B (D 2048, 28 blocks, 32 heads, 64x128 chunk, 4 374 image + 32 language condition tokens), pi_I at B=32, T=16,
DINOv2-base @224."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda:0"

RDT1B = dict(hidden=2048, depth=28, heads=32, horizon=64, action_dim=128, lang_token_dim=4096, img_token_dim=1152,
             state_token_dim=128, max_lang_cond_len=1024, img_cond_len=4374)


@pytest.fixture(scope="module")
def rdt1b():
    from models.rdt_runner import RDTRunner
    from vlatouch import synth
    cfg = {"rdt": {"hidden_size": 2048, "depth": 28, "num_heads": 32}, "lang_adaptor": "mlp2x_gelu", "img_adaptor": "mlp2x_gelu",
           "state_adaptor": "mlp3x_gelu",
           "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": 5, "beta_schedule": "squaredcos_cap_v2",
                               "prediction_type": "sample", "clip_sample": False}}
    r = RDTRunner(action_dim=128, pred_horizon=64, config=cfg, lang_token_dim=4096, img_token_dim=1152, state_token_dim=128,
                  max_lang_cond_len=1024, img_cond_len=4374, dtype=torch.bfloat16, device=DEV, init_weights=False)
    r.load_state_dict(synth.fill_state_dict_device(synth.rdt_runner_shapes(**RDT1B), torch.device(DEV), torch.bfloat16, seed=7), assign=True)
    return r


def rdt_inputs(B, L=32, seed=5):
    g = torch.Generator(device=DEV).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    amask = torch.zeros(B, 1, 128, device=DEV, dtype=torch.bfloat16)
    amask[:, :, :10] = 1.0
    return dict(lang=rn(B, L, 4096), mask=torch.ones(B, L, dtype=torch.bool, device=DEV), img=rn(B, 4374, 1152), state=rn(B, 1, 128),
                amask=amask, freq=torch.full((B,), 10.0, device=DEV), x0=rn(B, 64, 128).float())


def run(r, d, sl=slice(None)):
    return r.predict_action(d["lang"][sl], d["mask"][sl], d["img"][sl], d["state"][sl], d["amask"][sl], d["freq"][sl], x_init=d["x0"][sl]).float()


def test_rdt_1b_batch_invariance_determinism_and_mask(rdt1b):
    d = rdt_inputs(3)
    full = run(rdt1b, d)
    assert full.shape == (3, 64, 128) and torch.isfinite(full).all()
    assert torch.equal(full, run(rdt1b, d))                                   # deterministic
    assert float(full[:, :, 10:].abs().max()) == 0.0                          # action mask applied
    scale = float(full.abs().max())
    # every sample alone (different first/last-tile alignment of its keys in the tile stream: 4374 % 64 != 0) == in the batch.
    # Not bit-exact: the K/V projection tiles see different row neighbours only through fp32 accumulation order -> none; the
    # per-step GEMMs pick tile sizes from M, which changes the summation grouping of the 16-bit products.  Bar: the north-star tolerance on the
    # result (1e-2 of its scale; round 4 allowed twice that with bf16 activations), compared in the returned dtype (bf16: half an ulp at the scale is 2e-3 .. 4e-3)
    for b in range(3):
        alone = run(rdt1b, d, slice(b, b + 1))
        e = float((alone[0] - full[b]).abs().max())
        print(f"[RDT-1B batch invariance row {b}] {e:.3e} (scale {scale:.2f})")
        assert e <= 1e-2, (b, e, scale)                # FLAT (round 6): the north star's 1e-2 itself, not 1e-2 of the scale
    # language tokens under a False mask do not matter
    d2 = dict(d)
    d2["mask"] = d["mask"].clone()
    d2["mask"][:, 20:] = False
    a = run(rdt1b, d2)
    d3 = dict(d2)
    d3["lang"] = d2["lang"].clone()
    d3["lang"][:, 20:] = 7.0
    assert torch.equal(a, run(rdt1b, d3))
    assert float((a - full).abs().max()) > 0                                  # while unmasked ones do


def test_pi_refine_full_batch_split_invariance():
    """B=32, T=16, DINOv2-base @224: rows [0,16) and [16,32) refined separately == the batch of 32 (same noise rows)."""
    from residual_controller.bridge_controller import DiffusionController
    ctrl = cases.build_controller(DiffusionController, precision="bf16", device=DEV, size="base", stats_kind="nontrivial")
    g = np.random.default_rng(3)
    B, T = 32, 16
    mk = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(DEV)
    cam1, cam2 = mk(0.2 + 0.8 * g.random((B, 3, 224, 224))), mk(0.2 + 0.8 * g.random((B, 3, 224, 224)))
    state, forces, vla = mk(g.standard_normal((B, 10))), mk(g.standard_normal((B, 3))), mk(g.uniform(0, 1, (B, T, 10)))
    z = mk(g.standard_normal((10, B, T, 10)))
    full = ctrl.predict(state, vla, cam1, cam2, forces, noise=z)
    assert full.shape == (B, T, 10) and torch.isfinite(full).all()
    assert torch.equal(full, ctrl.predict(state, vla, cam1, cam2, forces, noise=z))
    for sl in (slice(0, 16), slice(16, 32)):
        part = ctrl.predict(state[sl], vla[sl], cam1[sl], cam2[sl], forces[sl], noise=z[:, sl].contiguous())
        e = float((part - full[sl]).abs().max())
        assert e < 5e-3, e          # tile choice depends on M (bf16/fp16 summation grouping); the target on a_hat is 1e-2


def test_two_streams_in_flight_are_bit_identical_to_one(rdt1b):
    """The throughput mode of bench.py keeps two batches in flight on two HIP streams, so kernels of different stages share CUs (an RDT-1B
    chunk beside the pi_I refinement: K|V projections / denoise-loop tiles beside DINOv2 tiles and the fused U-Net launches).  Every
    kernel must be indifferent to what runs beside it: the concurrent results equal the one-at-a-time results bit for bit, repeatedly.
    (Regression net for co-residency faults such as the packed-fp32 one found in vt_uconv.hip)"""
    from residual_controller.bridge_controller import DiffusionController
    ctrl = cases.build_controller(DiffusionController, precision="bf16", device=DEV, size="base", stats_kind="nontrivial")
    g = np.random.default_rng(17)
    B, T = 16, 16
    mk = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(DEV)
    cam1, cam2 = mk(0.2 + 0.8 * g.random((B, 3, 224, 224))), mk(0.2 + 0.8 * g.random((B, 3, 224, 224)))
    state, forces, vla = mk(g.standard_normal((B, 10))), mk(g.standard_normal((B, 3))), mk(g.uniform(0, 1, (B, T, 10)))
    z = mk(g.standard_normal((10, B, T, 10)))
    d = rdt_inputs(8, seed=9)
    ref_pi = ctrl.predict(state, vla, cam1, cam2, forces, noise=z)
    ref_rdt = run(rdt1b, d)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(s1):            # size the per-stream workspaces before the concurrent phase
        run(rdt1b, d)
    with torch.cuda.stream(s2):
        ctrl.predict(state, vla, cam1, cam2, forces, noise=z)
    torch.cuda.synchronize()
    for _ in range(3):
        outs_pi = []
        with torch.cuda.stream(s1):
            out_rdt = run(rdt1b, d)
        with torch.cuda.stream(s2):
            for _k in range(6):            # ~6 refinements fit beside one chunk generation
                outs_pi.append(ctrl.predict(state, vla, cam1, cam2, forces, noise=z))
        torch.cuda.synchronize()
        assert torch.equal(out_rdt, ref_rdt)
        for o in outs_pi:
            assert torch.equal(o, ref_pi)


def test_bench_pattern_full_graphs_in_flight_are_bit_identical_to_one(rdt1b):
    """bench.py's headline mode, exactly: three (round 4's default; two before) captured hipGraphs of the WHOLE step at B = 32 (RDT-1B chunk -> first 16 ticks x 10 EEF dims ->
    DINOv2-B x2 + MLP + 10-step SDE), one per HIP stream, replayed alternately so that two full steps are always in flight (RDT beside RDT,
    K|V projections beside denoise-loop tiles beside the fused U-Net launches).  With the start noise and the SDE noise held fixed every
    replay on either stream must reproduce the one-at-a-time result bit for bit."""
    from residual_controller.bridge_controller import DiffusionController
    from vlatouch import ops as _ops
    ctrl = cases.build_controller(DiffusionController, precision="bf16", device=DEV, size="base", stats_kind="nontrivial")
    g = np.random.default_rng(23)
    B, T = 32, 16
    mk = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(DEV)
    cam1, cam2 = mk(0.2 + 0.8 * g.random((B, 3, 224, 224))), mk(0.2 + 0.8 * g.random((B, 3, 224, 224)))
    state, forces = mk(g.standard_normal((B, 10))), mk(g.standard_normal((B, 3)))
    z = mk(g.standard_normal((10, B, T, 10)))
    d = rdt_inputs(B, seed=11)
    x0 = d["x0"]                                        # bf16-rounded values in fp32, what bench.py's device RNG hands over
    NS = 3
    vla_bufs = [torch.empty(B, T, 10, dtype=torch.float32, device=DEV) for _ in range(NS)]
    hold = [{} for _ in range(NS)]

    def step(si):
        chunk = rdt1b.predict_action(d["lang"], d["mask"], d["img"], d["state"], d["amask"], d["freq"], x_init=x0, return_fp32=True)
        vla = _ops.slice_cast(chunk, T, 10, out=vla_bufs[si])
        hold[si]["chunk"] = chunk
        hold[si]["out"] = ctrl.predict(state, vla, cam1, cam2, forces, noise=z)

    streams = [torch.cuda.Stream(device=DEV) for _ in range(NS)]
    with torch.cuda.stream(streams[0]):                 # the one-at-a-time reference (eager, nothing beside it)
        step(0)
        streams[0].synchronize()
        ref_chunk, ref_out = hold[0]["chunk"].clone(), hold[0]["out"].clone()
    assert torch.isfinite(ref_out).all() and torch.isfinite(ref_chunk).all()
    graphs = []
    for si, st in enumerate(streams):
        with torch.cuda.stream(st):
            step(si)                                     # sizes this stream's workspaces outside the capture
            st.synchronize()
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, stream=st):
                step(si)
            graphs.append(g_)
    torch.cuda.synchronize()
    for rnd in range(6):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                graphs[si].replay()
        # results of round `rnd` are read after both streams drain; the NEXT round is enqueued behind them, so two steps overlap fully
        torch.cuda.synchronize()
        for si in range(NS):
            assert torch.equal(hold[si]["chunk"], ref_chunk), (rnd, si, float((hold[si]["chunk"] - ref_chunk).abs().max()))
            assert torch.equal(hold[si]["out"], ref_out), (rnd, si, float((hold[si]["out"] - ref_out).abs().max()))
    # and back to back without a host sync in between (the bench's steady state: step i+2 queued behind step i on the same stream)
    for rnd in range(4):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                graphs[si].replay()
    torch.cuda.synchronize()
    for si in range(NS):
        assert torch.equal(hold[si]["chunk"], ref_chunk) and torch.equal(hold[si]["out"], ref_out)


# ---------------------------------------------------------------- full-size parity against the oracle
_ORACLE_CACHE = {}


def _oracle_episode(r, d, b, steps):
    """The oracle's predict_action (fp32 math) on episode b alone, with the runner's bf16-rounded weights, inputs and start noise.  Every runner of this file
    carries the same seed-7 weights, so a result is reused for the same inputs (keyed by a checksum of the episode's inputs): the oracle costs ~10 s per episode."""
    key = (b, steps, r.rms_mode, float(d["x0"][b].double().sum()), float(d["img"][b].double().sum()), float(d["lang"][b].double().sum()))
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = _oracle_episode_uncached(r, d, b, steps)
    return _ORACLE_CACHE[key]


def _oracle_episode_uncached(r, d, b, steps):
    from oracle import rdt as orr
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    sd = {k: v.float().cpu() for k, v in r.state_dict().items()}
    one = lambda t: (t[b:b + 1].float().cpu() if t.dtype != torch.bool else t[b:b + 1].cpu())
    return orr.predict_action(sd, one(d["lang"]), one(d["mask"]), one(d["img"]), one(d["state"]), one(d["amask"]), one(d["freq"]), one(d["x0"]),
                              heads=32, horizon=64, num_inference_steps=steps, rms_mode=r.rms_mode)[0]


def test_rdt_1b_batch32_rows_vs_oracle(rdt1b):
    """BASELINE configs[3]'s RDT leg exactly as bench.py times it: B=32 (M = 2144 rows / 139 968 condition rows -> gemm_ppk_kernel,
    gemm_pp256_kernel<..,1|2> and the un-split attn_kvt_kernel), 5 DPM-Solver++ steps; rows 0 and 31 against the oracle run on those
    single episodes.  Bar: the north star's FLAT 1e-2 on the chunk (round 6; the product default = fp16 activations under the range guard; models/rdt_runner.py:122-165,225-250)."""
    d = rdt_inputs(32, seed=17)
    rdt1b.num_inference_timesteps = 5
    full = run(rdt1b, d)
    assert full.shape == (32, 64, 128) and torch.isfinite(full).all()
    for b in (0, 31):
        ref = _oracle_episode(rdt1b, d, b, 5)
        scale = float(ref.abs().max())
        e = float((full[b].cpu() - ref).abs().max())
        print(f"[RDT-1B B=32 row {b}] scale {scale:.3f}  |hip16 - oracle32| {e:.3e}  ({e / scale:.2e} of scale)")
        assert e <= 1e-2, (b, e, scale)                # FLAT 1e-2 (round 6; measured 4.3e-3 in the returned bf16, 8e-4 on the fp32 hand-over)


def test_rdt_1b_batch32_activation_types_and_rmsnorm_forms_vs_oracle():
    """The same B = 32 configuration (M = 2144 rows: weights-in-registers tiles with the RMSNorm hand-off, persistent K|V projection, cached
    cross-attention) in the OTHER settings a deployment can ask for, row 0 against the oracle: the reference's own execution dtype for the activations
    (compute_dtype="bf16": within 1e-2 of the output scale; the fp16 default at least 3x closer on the same inputs), and timm <= 1.0.8's variance
    RmsNorm (what released RDT-1B checkpoints need: hand-off with row sums, online softmax) with fp16 activations."""
    from models.rdt_runner import RDTRunner
    from vlatouch import synth
    d = rdt_inputs(32, seed=17)
    errs, scales = {}, {}
    refs = {}
    for compute, rms in (("bf16", "meansq"), ("f16", "meansq"), ("f16", "var")):
        cfg = {"rdt": {"hidden_size": 2048, "depth": 28, "num_heads": 32, "rms_norm": rms}, "lang_adaptor": "mlp2x_gelu", "img_adaptor": "mlp2x_gelu",
               "state_adaptor": "mlp3x_gelu",
               "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": 5, "beta_schedule": "squaredcos_cap_v2",
                                   "prediction_type": "sample", "clip_sample": False}}
        r = RDTRunner(action_dim=128, pred_horizon=64, config=cfg, lang_token_dim=4096, img_token_dim=1152, state_token_dim=128, max_lang_cond_len=1024,
                      img_cond_len=4374, dtype=torch.bfloat16, device=DEV, init_weights=False, compute_dtype=compute)
        r.load_state_dict(synth.fill_state_dict_device(synth.rdt_runner_shapes(**RDT1B), torch.device(DEV), torch.bfloat16, seed=7), assign=True)
        out = r.predict_action(d["lang"], d["mask"], d["img"], d["state"], d["amask"], d["freq"], x_init=d["x0"], return_fp32=True)
        if rms not in refs:
            refs[rms] = _oracle_episode(r, d, 0, 5)
        errs[(compute, rms)] = float((out[0].cpu() - refs[rms]).abs().max())
        scales[rms] = float(refs[rms].abs().max())
        del r, out
        torch.cuda.empty_cache()
    print("[RDT-1B B=32 row 0] |hip - oracle32|: " + ", ".join(f"{c} activations / {m}: {e:.3e} (scale {scales[m]:.2f})" for (c, m), e in errs.items()))
    s_ms, s_var = max(1.0, scales["meansq"]), max(1.0, scales["var"])
    assert errs[("bf16", "meansq")] <= 1e-2 * s_ms and errs[("f16", "meansq")] <= 2.5e-3 and 3 * errs[("f16", "meansq")] <= errs[("bf16", "meansq")], errs      # fp16: flat
    assert errs[("f16", "var")] <= 2.5e-3, errs


@pytest.mark.slow
def test_rdt_1b_50_steps_batch16(rdt1b):
    """BASELINE configs[2]: RDT-1B, 50 denoise steps, B=16 — determinism, batch invariance and row-0 parity against the oracle."""
    d = rdt_inputs(16, seed=23)
    rdt1b.num_inference_timesteps = 50
    try:
        full = run(rdt1b, d)
        assert full.shape == (16, 64, 128) and torch.isfinite(full).all()
        assert torch.equal(full, run(rdt1b, d))
        scale = float(full.abs().max())
        alone = run(rdt1b, d, slice(5, 6))
        e_inv = float((alone[0] - full[5]).abs().max())
        print(f"[RDT-1B 50 steps] scale {scale:.3f}  batch-invariance row 5: {e_inv:.3e}")
        assert e_inv <= 1e-2, (e_inv, scale)
        ref = _oracle_episode(rdt1b, d, 0, 50)
        e = float((full[0].cpu() - ref).abs().max())
        rs = float(ref.abs().max())
        print(f"[RDT-1B B=16 50 steps row 0] scale {rs:.3f}  |hip16 - oracle32| {e:.3e}  ({e / rs:.2e} of scale)")
        assert e <= 1e-2, (e, rs)
    finally:
        rdt1b.num_inference_timesteps = 5


def test_rdt_chunk_feeds_pi_refine_vs_oracle():
    """The chained path of frank_inference_eef.py:495-533 at a size the oracle runs in seconds: RDT chunk -> first T ticks x 10 EEF
    dims -> DiffusionController.predict, fp32 end to end, against oracle(RDT) -> oracle(predict)."""
    from oracle import rdt as orr
    from oracle import controller as oc
    from tests.test_gpu_rdt import make_runner
    from residual_controller.bridge_controller import DiffusionController
    cfg = cases.RDT_TINY
    r = make_runner(cfg, torch.float32)
    ri = cases.rdt_inputs(cfg, 2, 12)
    T = cfg["horizon"]                                              # 8 ticks (divisible by 4 for the U-Net)
    chunk = r.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"],
                             x_init=ri["x_init"])
    inp = cases.predict_inputs(2, T, 224)
    ctrl = cases.build_controller(DiffusionController, precision="fp32", device=DEV)
    out = ctrl.predict(inp["state"], chunk[:, :T, :10].float(), inp["cam1"], inp["cam2"], inp["forces"], noise=inp["z"])
    ref_chunk = orr.predict_action(cases.rdt_sd(cfg), ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"],
                                   ri["freq"], ri["x_init"], heads=cfg["heads"], horizon=cfg["horizon"], num_inference_steps=5)
    ref = oc.predict(cases.dino_sd("small"), 6, cases.state_encoder_sd(781), cases.si_net_sd("ema"), cases.stats("nontrivial"),
                     inp["state"], ref_chunk[:, :T, :10], inp["cam1"], inp["cam2"], inp["forces"], inp["z"])
    e = float((out.cpu() - ref).abs().max())
    print(f"[chained RDT -> pi_I fp32] max|a_hat - oracle| = {e:.3e}")
    assert out.shape == (2, T, 10) and e < 1e-4, e


def _rdt1b_runner(compute, rms):
    """A second RDT-1B runner on the seed-7 weights of the `rdt1b` fixture with another activation type / RmsNorm form (2.4 GB + packed copies: callers drop it)."""
    from models.rdt_runner import RDTRunner
    from vlatouch import synth
    cfg = {"rdt": {"hidden_size": 2048, "depth": 28, "num_heads": 32, "rms_norm": rms}, "lang_adaptor": "mlp2x_gelu", "img_adaptor": "mlp2x_gelu",
           "state_adaptor": "mlp3x_gelu",
           "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": 5, "beta_schedule": "squaredcos_cap_v2",
                               "prediction_type": "sample", "clip_sample": False}}
    r = RDTRunner(action_dim=128, pred_horizon=64, config=cfg, lang_token_dim=4096, img_token_dim=1152, state_token_dim=128, max_lang_cond_len=1024,
                  img_cond_len=4374, dtype=torch.bfloat16, device=DEV, init_weights=False, compute_dtype=compute)
    r.load_state_dict(synth.fill_state_dict_device(synth.rdt_runner_shapes(**RDT1B), torch.device(DEV), torch.bfloat16, seed=7), assign=True)
    return r


def _chain(r, d, force_dim, rows, kinds, tag, bf16_state_too=False):
    """BASELINE configs[3] END TO END in the low-precision mode at B = 32: runner `r`'s RDT-1B chunk (5-step DPM-Solver++) -> first 16 ticks x 10 EEF dims
    (`slice_cast`) -> DiffusionController.predict (DINOv2-base, T = 16, injected SDE noise) = a_hat, exactly as bench.py's step chains them
    (frank_inference_eef.py:495-533, bridge_controller.py:149-182), against oracle.rdt.predict_action -> oracle.controller.predict (fp32 math on the same
    bf16-rounded RDT weights / inputs / start noise, the runner's RmsNorm form) on the episodes `rows`.  Returns ({stats kind: worst |a_hat - oracle|}, bars)."""
    from oracle import controller as oc
    from residual_controller.bridge_controller import DiffusionController
    from vlatouch import ops as _ops, _lib as L
    B, T = 32, 16
    r.num_inference_timesteps = 5
    g = np.random.default_rng(31 + force_dim)
    mk = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))
    cam1, cam2 = mk(0.2 + 0.8 * g.random((B, 3, 224, 224))), mk(0.6 * g.random((B, 3, 224, 224)))
    state, forces = mk(g.standard_normal((B, 10))), mk(g.standard_normal((B, force_dim)))
    z = mk(g.standard_normal((10, B, T, 10)))
    ctrl = cases.build_controller(DiffusionController, precision="bf16", device=DEV, size="base", stats_kind="nontrivial", force_dim=force_dim)
    predict_chunk = lambda: r.predict_action(d["lang"], d["mask"], d["img"], d["state"], d["amask"], d["freq"], x_init=d["x0"], return_fp32=True)
    chunk = predict_chunk()
    chunk_bf16_state = None
    if bf16_state_too:
        eng = r.engine()
        try:      # the reference's own rounding points (bf16 solver state, rdt_runner.py:160), for the record
            L.check(L.lib().vt_rdt_set_state_precision(eng._h, 0), "state precision")
            chunk_bf16_state = predict_chunk().float().cpu()
        finally:
            L.lib().vt_rdt_set_state_precision(eng._h, int(eng.solver_state == "fp32"))
    vla = _ops.slice_cast(chunk, T, 10)
    assert vla.shape == (B, T, 10) and vla.dtype == torch.float32
    dev = lambda t: t.to(DEV)
    nt = cases.stats("nontrivial")
    lo, hi = vla.amin(dim=(0, 1)).cpu(), vla.amax(dim=(0, 1)).cpu()
    covering = dict(nt, vla_mins=lo, vla_maxs=hi, vla_range=hi - lo)
    stats = {k: v for k, v in {"unit": cases.stats("unit"), "covering": covering, "narrow": nt}.items() if k in kinds}
    gain = float((nt["action_range"][:9] / nt["vla_range"][:9]).max())                 # dim 9 of the fixture has a zero vla range (the < 1e-6 guard)
    bars = {"unit": 1e-2, "covering": 1e-2, "narrow": 1e-2 * gain + 2e-3}
    got = {}
    for kind, st in stats.items():
        ctrl.stats = {k: v.to(DEV) for k, v in st.items()}
        got[kind] = ctrl.predict(dev(state), vla, dev(cam1), dev(cam2), dev(forces), noise=dev(z)).cpu()
        assert got[kind].shape == (B, T, 10) and torch.isfinite(got[kind]).all()
    sds = (cases.dino_sd("base"), cases.state_encoder_sd(2 * 768 + 10 + force_dim), cases.si_net_sd("ema"))
    worst = {k: 0.0 for k in stats}
    for b in rows:
        ref_chunk = _oracle_episode(r, d, b, 5)                              # [64, 128] fp32
        e_chunk = float((chunk[b].float().cpu() - ref_chunk).abs().max())
        extra = ""
        if chunk_bf16_state is not None:
            extra = f", {float((chunk_bf16_state[b] - ref_chunk).abs().max()):.3e} with the reference's bf16 rounding points"
        print(f"[chain {tag} B=32 force_dim {force_dim} row {b}] chunk err {e_chunk:.3e} with the fp32 solver state{extra} (chunk scale {float(ref_chunk.abs().max()):.2f})")
        one = slice(b, b + 1)
        for kind, st in stats.items():
            ref = oc.predict(sds[0], 12, sds[1], sds[2], st, state[one], ref_chunk[None, :T, :10], cam1[one], cam2[one], forces[one], z[:, one])
            # the same controller fed the ORACLE's chunk: what the pi_I leg alone contributes
            ctrl.stats = {k: v.to(DEV) for k, v in st.items()}
            vla_o = vla.clone()
            vla_o[b] = ref_chunk[:T, :10].to(DEV)
            pi_only = ctrl.predict(dev(state), vla_o, dev(cam1), dev(cam2), dev(forces), noise=dev(z)).cpu()
            e_pi = float((pi_only[b] - ref[0]).abs().max())
            e = float((got[kind][b] - ref[0]).abs().max())
            worst[kind] = max(worst[kind], e)
            print(f"[chain {tag} B=32 force_dim {force_dim} row {b} stats {kind}] pi_I alone {e_pi:.3e}  a_hat chained {e:.3e}  (a_hat scale "
                  f"{float(ref.abs().max()):.2f}, bar {bars[kind]:.2e})")
    return worst, bars


@pytest.mark.parametrize("force_dim", [3, 64])
def test_chained_rdt1b_bf16_chunk_feeds_pi_refine_batch32_vs_oracle(rdt1b, force_dim):
    """The quantity the north-star tolerance is stated on (a_hat of the chained path, `_chain`), the product default (compute_dtype="auto" = IEEE fp16 activations
    under the range guard, mean-square RmsNorm), episodes 0 and 31.  force_dim = 64: the 64-d tactile vector of BASELINE.json's workload (bridge_controller.py:25).

    Normalisation statistics (controller_dataset.py:222-229 computes them FROM the data they normalise):
      * "unit"     = the bench's: a_hat in the chunk's own units;                                              FLAT bar 1e-2 on a_hat
      * "covering" = non-trivial: vla_mins / vla_maxs = the per-dimension range of this batch's chunks (what a dataset statistic is), expert
                     action range from the non-trivial fixture (0.5 .. 2.0 wide, offset);                      FLAT bar 1e-2 on a_hat
      * "narrow"   = the fixture's own vla range (0.4 .. 2.1 wide around -0.3), which does NOT cover an RDT-1B chunk of scale 1.7: |x_n| reaches 6 and
                     the reference's own arithmetic multiplies any chunk error by action_range / vla_range (up to 3.0) on the way to a_hat, whose
                     scale is then ~8 — beyond what bf16 resolves to 1e-2 (one bf16 ulp at 8 is 3e-2).  Bar: 1e-2 x that gain (+ the pi_I leg)."""
    d = rdt_inputs(32, seed=29)
    worst, bars = _chain(rdt1b, d, force_dim, (0, 31), ("unit", "covering", "narrow"), "f16 act / meansq", bf16_state_too=True)
    assert rdt1b.compute_dtype == torch.float16 and rdt1b.overflowed() == 0        # the range guard stayed clean: "auto" kept fp16
    for kind in worst:
        assert worst[kind] <= bars[kind], (kind, worst[kind], bars[kind])


def test_chained_rdt1b_var_rmsnorm_vs_oracle():
    """The chained a_hat in the arithmetic released RDT-1B checkpoints need: timm==1.0.3's variance `RmsNorm` (models/rdt/blocks.py:22,150-156: hand-off with row
    sums, online softmax in the cached cross-attention), fp16 activations.  Inputs = seed 17, episode 0: the oracle episode of
    test_rdt_1b_batch32_activation_types_and_rmsnorm_forms_vs_oracle is reused (_ORACLE_CACHE), so this leg costs the pi_I oracle only.  FLAT bar 1e-2."""
    r = _rdt1b_runner("auto", "var")
    try:
        d = rdt_inputs(32, seed=17)
        worst, bars = _chain(r, d, 3, (0,), ("unit", "covering"), "f16 act / var")
        assert r.compute_dtype == torch.float16 and r.overflowed() == 0
        for kind in worst:
            assert worst[kind] <= 1e-2, (kind, worst[kind])
    finally:
        del r
        torch.cuda.empty_cache()


BF16_CHAIN = {}


def test_chained_rdt1b_bf16_activations_measured():
    """The same chain with the reference's own execution dtype for the activations (compute_dtype="bf16"), seed 29 / episodes 0 and 31 (oracle episodes cached by
    the default-dtype test above): measured and recorded for the strict-xfail test below; held only to 3e-2 here (round 6 measured 1.7e-2 and 2.1e-2 on the covering statistics with two
    different — equally valid — score bounds of the fixed-maximum softmax: bf16 probabilities round differently for every bound)."""
    r = _rdt1b_runner("bf16", "meansq")
    try:
        d = rdt_inputs(32, seed=29)
        worst, _ = _chain(r, d, 3, (0, 31), ("unit", "covering"), "bf16 act / meansq")
        BF16_CHAIN.update(worst)
        for kind in worst:
            assert worst[kind] <= 3e-2, (kind, worst[kind])
    finally:
        del r
        torch.cuda.empty_cache()


@pytest.mark.xfail(strict=True, reason="bf16 ACTIVATIONS miss the north star's flat 1e-2 on the chained a_hat at RDT-1B (28 blocks x 5 steps of bf16 rounding: 0.9 - 1.1e-2 with unit statistics, 1.7 - 2.1e-2 with data-covering ones) — "
                                       "the reason the product's default computes in IEEE fp16 under the range guard (DESIGN.md section 3); if this ever passes, revisit the default")
def test_chained_rdt1b_bf16_activations_meet_the_flat_bar():
    assert BF16_CHAIN, "test_chained_rdt1b_bf16_activations_measured must run first (same module, collected above)"
    assert max(BF16_CHAIN.values()) <= 1e-2, BF16_CHAIN
