"""GPU parity for the RDT path: the mirrors in vla-touch_amd/models against
  G8 — outputs of the reference's own models/rdt/model.py (imported with the timm shim) in fp32 and bf16,
  G9 — the oracle's predict_action (DPM-Solver++ restated; parity UNPINNED, see oracle/dpm_solver.py),
and against the oracle run live.  bf16 bar: the HIP path must be as close to the exact (fp32) result as the
reference's own bf16 execution is (within 1.5x), and within 1e-2 of output scale where that is achievable."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def G(name):
    return np.load(f"{cases.GOLDEN}/{name}.npz")


def err(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max())


def make_rdt(cfg, dtype):
    from models.rdt.model import RDT
    m = RDT(output_dim=cfg["action_dim"], horizon=cfg["horizon"], hidden_size=cfg["hidden"], depth=cfg["depth"], num_heads=cfg["heads"],
            max_lang_cond_len=cfg["max_lang_cond_len"], img_cond_len=cfg["img_cond_len"], dtype=dtype)
    sd = cases.rdt_sd(cfg, torch.float32)
    m.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")})
    return m


RUNNER_CFG = {"rdt": None, "lang_adaptor": "mlp2x_gelu", "img_adaptor": "mlp2x_gelu", "state_adaptor": "mlp3x_gelu",
              "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": 5, "beta_schedule": "squaredcos_cap_v2",
                                  "prediction_type": "sample", "clip_sample": False}}


def make_runner(cfg, dtype, compute=None, **over):
    """compute: the 16-bit activation type of a bf16 runner (None = "auto": fp16 under the range guard; "f16" / "bf16" pin it; "bf16" = the reference's execution dtype)."""
    from models.rdt_runner import RDTRunner
    c = dict(RUNNER_CFG)
    c["rdt"] = {"hidden_size": cfg["hidden"], "depth": cfg["depth"], "num_heads": cfg["heads"]}
    c.update(over)
    r = RDTRunner(action_dim=cfg["action_dim"], pred_horizon=cfg["horizon"], config=c, lang_token_dim=cfg["lang_token_dim"],
                  img_token_dim=cfg["img_token_dim"], state_token_dim=cfg["state_token_dim"], max_lang_cond_len=cfg["max_lang_cond_len"],
                  img_cond_len=cfg["img_cond_len"], dtype=dtype, device="cuda:0", compute_dtype=compute)
    r.load_state_dict(cases.rdt_sd(cfg, torch.float32))
    return r


@pytest.mark.parametrize("tag,cfg,B,L", [("tiny", cases.RDT_TINY, 2, 12), ("wide", cases.RDT_WIDE, 1, 20)])
def test_rdt_forward_fp32_golden(tag, cfg, B, L):
    g = G("g8_rdt_fwd")[f"{tag}_f32"]
    m = make_rdt(cfg, torch.float32)
    ri = cases.rdt_inputs(cfg, B, L)
    y = m(ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"])
    assert y.shape == g.shape
    e = err(y, g)
    assert e < 2e-4 * max(1.0, float(np.abs(g).max())), (tag, e)


@pytest.mark.parametrize("tag,cfg,B,L", [("tiny", cases.RDT_TINY, 2, 12), ("wide", cases.RDT_WIDE, 1, 20)])
def test_rdt_forward_bf16_vs_reference_bf16(tag, cfg, B, L):
    g = G("g8_rdt_fwd")
    exact, ref16 = g[f"{tag}_f32"], g[f"{tag}_bf16"]
    m = make_rdt(cfg, torch.bfloat16)
    ri = cases.rdt_inputs(cfg, B, L, dtype=torch.bfloat16)
    y = m(ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"])
    assert y.dtype == torch.bfloat16
    scale = float(np.abs(exact).max())
    e_hip, e_ref, e_pair = err(y, exact), err(ref16, exact), err(y, ref16)
    print(f"[{tag}] scale {scale:.2f}: |hip16-exact| {e_hip:.3e}  |ref16-exact| {e_ref:.3e}  |hip16-ref16| {e_pair:.3e}")
    assert e_hip <= max(1e-2 * scale, 1.5 * e_ref), (tag, e_hip, e_ref)


def test_rdt_forward_per_sample_timesteps_and_var_rmsnorm():
    from oracle import rdt as orr
    from models.rdt.model import RDT
    cfg = cases.RDT_TINY
    sd = cases.rdt_sd(cfg, torch.float32)
    ri = cases.rdt_inputs(cfg, 2, 12)
    t = torch.tensor([437.0, 12.0])
    for mode in ("meansq", "var"):
        m = RDT(output_dim=cfg["action_dim"], horizon=cfg["horizon"], hidden_size=cfg["hidden"], depth=cfg["depth"], num_heads=cfg["heads"],
                max_lang_cond_len=cfg["max_lang_cond_len"], img_cond_len=cfg["img_cond_len"], dtype=torch.float32, rms_mode=mode)
        m.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")})
        ref = orr.rdt_forward(sd, ri["x"], ri["freq"], t, ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"], heads=cfg["heads"],
                              horizon=cfg["horizon"], rms_mode=mode)
        y = m(ri["x"], ri["freq"], t, ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"])
        assert err(y, ref.numpy()) < 2e-4 * max(1.0, float(ref.abs().max())), mode


@pytest.mark.parametrize("dname,dtype,compute", [("f32", torch.float32, None), ("bf16", torch.bfloat16, "bf16"), ("bf16", torch.bfloat16, "f16")])
def test_predict_action_vs_oracle_sampler(dname, dtype, compute):
    """bf16 model, both 16-bit activation types: "bf16" = the reference's execution dtype (held to the reference's own bf16 error), "f16" = the
    default (bf16 weights converted exactly, IEEE fp16 activations: held to a 4x tighter bar)."""
    g = G(f"g9_rdt_sample_{dname}_UNPINNED")
    exact = G("g9_rdt_sample_f32_UNPINNED")["out"]
    cfg = cases.RDT_TINY
    r = make_runner(cfg, dtype, compute=compute)
    assert r.compute_dtype == {None: torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[compute]
    ri = cases.rdt_inputs(cfg, 2, 12, dtype=dtype)
    out = r.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"],
                           x_init=ri["x_init"])
    assert out.shape == (2, cfg["horizon"], cfg["action_dim"]) and out.dtype == dtype
    assert float(out[..., 10:].abs().max()) == 0.0                       # action mask applied (rdt_runner.py:163)
    scale = float(np.abs(exact).max())
    if dname == "f32":
        assert err(out, g["out"]) < 2e-4 * max(1.0, scale), err(out, g["out"])
    else:
        e_hip, e_ref = err(out, exact), err(g["out"], exact)
        print(f"[compute {compute}] scale {scale:.2f}: |hip16-exact| {e_hip:.3e}  |oracle16-exact| {e_ref:.3e}")
        assert e_hip <= max(1e-2 * scale, 1.5 * e_ref), (e_hip, e_ref)
        if compute == "f16":      # the result is returned in bf16 (the model's dtype): half a bf16 ulp of the scale is the floor
            out32 = r.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"],
                                     x_init=ri["x_init"], return_fp32=True)
            e32 = err(out32, exact)
            print(f"[compute f16, fp32 hand-over] |hip16-exact| {e32:.3e}")
            assert e32 <= 4e-3 * scale, (e32, scale)              # (the tiny config: D = 256 averages less than RDT-1B, where it is 7e-4 of the scale)


def test_predict_action_reference_rounding_points_vs_bf16_golden():
    """solver_state="bf16" + compute_dtype="bf16" reproduces the REFERENCE's own rounding points (the start noise drawn in bf16, `noisy_action.to(dtype)` after
    every scheduler step, bf16 model output; models/rdt_runner.py:137-139,160): held to the oracle run in bf16 arithmetic (g9 ..._bf16, the reference-dtype golden)
    so that the mode does not rot behind the fp32-state default.  Two bf16 executions of 4 blocks x 5 steps differ by rounding, not by bits: the bar is
    their common distance to the fp32 result."""
    g16 = G("g9_rdt_sample_bf16_UNPINNED")["out"]
    exact = G("g9_rdt_sample_f32_UNPINNED")["out"]
    cfg = cases.RDT_TINY
    ri = cases.rdt_inputs(cfg, 2, 12, dtype=torch.bfloat16)
    args = (ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"])
    from models.rdt_runner import RDTRunner
    c = dict(RUNNER_CFG)
    c["rdt"] = {"hidden_size": cfg["hidden"], "depth": cfg["depth"], "num_heads": cfg["heads"]}
    outs = {}
    for state in ("bf16", "fp32"):
        r = RDTRunner(action_dim=cfg["action_dim"], pred_horizon=cfg["horizon"], config=c, lang_token_dim=cfg["lang_token_dim"], img_token_dim=cfg["img_token_dim"],
                      state_token_dim=cfg["state_token_dim"], max_lang_cond_len=cfg["max_lang_cond_len"], img_cond_len=cfg["img_cond_len"], dtype=torch.bfloat16,
                      device="cuda:0", compute_dtype="bf16", solver_state=state)
        r.load_state_dict(cases.rdt_sd(cfg, torch.float32))
        assert r.engine().solver_state == state
        outs[state] = r.predict_action(*args, x_init=ri["x_init"], return_fp32=True)
    scale = float(np.abs(exact).max())
    o16 = outs["bf16"]
    assert torch.equal(o16, o16.to(torch.bfloat16).float())                  # every value sits on the bf16 grid (the state is rounded after each step, the mask applied in bf16)
    assert not torch.equal(outs["fp32"], outs["fp32"].to(torch.bfloat16).float())
    e_mode, e_ref, e_pair = err(o16, exact), err(g16, exact), err(o16, g16)
    print(f"[solver_state bf16] scale {scale:.2f}: |hip - exact| {e_mode:.3e}  |oracle16 - exact| {e_ref:.3e}  |hip - oracle16| {e_pair:.3e}  (fp32 state: {err(outs['fp32'], exact):.3e})")
    assert e_mode <= max(1e-2 * scale, 1.5 * e_ref), (e_mode, e_ref)
    assert e_pair <= e_mode + e_ref + 1e-6


def test_conditional_sample_equals_predict_action_and_errors():
    from models.rdt_runner import RDTRunner
    cfg = cases.RDT_TINY
    r = make_runner(cfg, torch.float32)
    ri = cases.rdt_inputs(cfg, 2, 12)
    full = r.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"], x_init=ri["x_init"])
    st = torch.cat([ri["state_tokens"], ri["action_mask"]], dim=2)
    lang_c, img_c, state_traj = r.adapt_conditions(ri["lang_tokens"], ri["img_tokens"], st)
    assert lang_c.shape == (2, 12, cfg["hidden"]) and state_traj.shape == (2, 1, cfg["hidden"])
    two = r.conditional_sample(lang_c, ri["lang_mask"], img_c, state_traj, ri["action_mask"], ri["freq"], x_init=ri["x_init"])
    assert err(two, full.cpu().numpy()) < 1e-4
    # linear adaptors + epsilon prediction + a different step count run through the same driver
    r2 = make_runner(dict(cfg), torch.float32)
    r2.num_inference_timesteps = 3
    r2.prediction_type = "epsilon"
    out = r2.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"], x_init=ri["x_init"])
    assert torch.isfinite(out).all()
    from oracle import rdt as orr
    ref = orr.predict_action(cases.rdt_sd(cfg), ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"],
                             ri["freq"], ri["x_init"], heads=cfg["heads"], horizon=cfg["horizon"], num_inference_steps=3, prediction_type="epsilon")
    assert err(out, ref.numpy()) < 5e-4 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError):
        r.build_condition_adapter("conv3x", 8, 8)
    r.prediction_type = "v"
    with pytest.raises(ValueError):
        r.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"])
    r.prediction_type = "sample"


def test_rdt_wide_batch_takes_large_gemm_path_with_fused_headnorm():
    """D=2048, B=4 (M = 268 rows): the GEMMs run on the LDS-DMA kernel with q/k RMSNorm fused in the epilogue."""
    from oracle import rdt as orr
    cfg = cases.RDT_WIDE
    g = G("g8_rdt_fwd")
    m = make_rdt(cfg, torch.bfloat16)
    ri = cases.rdt_inputs(cfg, 4, 20, seed=9, dtype=torch.bfloat16)
    y = m(ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"])
    sd = cases.rdt_sd(cfg, torch.bfloat16)
    f32 = {k: v.float() for k, v in sd.items()}
    rf = {k: (v.float() if v.is_floating_point() else v) for k, v in ri.items()}
    exact = orr.rdt_forward(f32, rf["x"], rf["freq"], rf["t"], rf["lang_c"], rf["img_c"], lang_mask=rf["lang_mask"], heads=cfg["heads"], horizon=cfg["horizon"])
    ref16 = orr.rdt_forward(sd, ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"], heads=cfg["heads"], horizon=cfg["horizon"])
    scale = float(exact.abs().max())
    e_hip, e_ref = err(y, exact.numpy()), err(ref16.float(), exact.numpy())
    print(f"[wide B4] scale {scale:.2f}: |hip16-exact| {e_hip:.3e}  |ref16-exact| {e_ref:.3e}")
    assert e_hip <= max(1e-2 * scale, 1.5 * e_ref), (e_hip, e_ref)


@pytest.mark.parametrize("img_len", [70, 75, 128])
def test_rdt_condition_cache_written_by_gemm_epilogues(img_len):
    """B=12 makes the condition K/V projections (M = 12*img_len rows) take the large-GEMM path, whose epilogues write the
    per-(batch, head) [K | Vt] tile stream directly: even / odd lengths exercise the aligned and the element-wise Vt stores,
    partial last tiles and 4-key groups that straddle a batch or tile boundary."""
    from oracle import rdt as orr
    cfg = dict(cases.RDT_WIDE, img_cond_len=img_len)
    m = make_rdt(cfg, torch.bfloat16)
    ri = cases.rdt_inputs(cfg, 12, 20, seed=11, dtype=torch.bfloat16)
    y = m(ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"])
    sd = cases.rdt_sd(cfg, torch.bfloat16)
    f32 = {k: v.float() for k, v in sd.items()}
    rf = {k: (v.float() if v.is_floating_point() else v) for k, v in ri.items()}
    exact = orr.rdt_forward(f32, rf["x"], rf["freq"], rf["t"], rf["lang_c"], rf["img_c"], lang_mask=rf["lang_mask"], heads=cfg["heads"], horizon=cfg["horizon"])
    ref16 = orr.rdt_forward(sd, ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"], heads=cfg["heads"], horizon=cfg["horizon"])
    scale = float(exact.abs().max())
    e_hip, e_ref = err(y, exact.numpy()), err(ref16.float(), exact.numpy())
    print(f"[img_len {img_len}] scale {scale:.2f}: |hip16-exact| {e_hip:.3e}  |ref16-exact| {e_ref:.3e}")
    assert torch.isfinite(y.float()).all()
    assert e_hip <= max(1e-2 * scale, 1.5 * e_ref), (e_hip, e_ref)


@pytest.mark.parametrize("dname,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("B,L,valid", [(1, 1, 1), (3, 16, 1), (2, 16, 16), (5, 7, 4)])
def test_rdt_forward_edge_language_lengths(dname, dtype, B, L, valid):
    """Language conditions at the edges, against the oracle run live: a single token, the maximum length, samples with only
    `valid` unmasked tokens (the reference passes -inf masks to SDPA, blocks.py:116-123), odd batch sizes."""
    from oracle import rdt as orr
    cfg = cases.RDT_TINY
    m = make_rdt(cfg, dtype)
    ri = cases.rdt_inputs(cfg, B, L, seed=20 + B, dtype=dtype)
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[:, :valid] = True
    mask[-1, :] = True                                     # last sample fully valid
    ri["lang_mask"] = mask
    y = m(ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"])
    sd = cases.rdt_sd(cfg, torch.float32)
    rf = {k: (v.float() if v.is_floating_point() else v) for k, v in ri.items()}
    exact = orr.rdt_forward(sd, rf["x"], rf["freq"], rf["t"], rf["lang_c"], rf["img_c"], lang_mask=rf["lang_mask"], heads=cfg["heads"],
                            horizon=cfg["horizon"])
    assert y.shape == exact.shape and torch.isfinite(y.float()).all()
    scale = max(1.0, float(exact.abs().max()))
    e = err(y, exact.numpy())
    assert e < (2e-4 if dname == "f32" else 3e-2) * scale, (dname, B, L, valid, e)


@pytest.mark.parametrize("dname,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_rdt_wide_var_rmsnorm(dname, dtype):
    """timm==1.0.3 `RmsNorm` (rsqrt(var_unbiased + eps), the upstream-checkpoint setting; models/rdt/blocks.py:22) at D=2048, B=4:
    row norms, the q/k head norms fused into the large-GEMM epilogues and the cached-condition k_norm all take `var` mode."""
    from oracle import rdt as orr
    from models.rdt.model import RDT
    cfg = cases.RDT_WIDE
    sd16 = cases.rdt_sd(cfg, dtype)
    m = RDT(output_dim=cfg["action_dim"], horizon=cfg["horizon"], hidden_size=cfg["hidden"], depth=cfg["depth"], num_heads=cfg["heads"],
            max_lang_cond_len=cfg["max_lang_cond_len"], img_cond_len=cfg["img_cond_len"], dtype=dtype, rms_mode="var")
    m.load_state_dict({k[len("model."):]: v for k, v in cases.rdt_sd(cfg, torch.float32).items() if k.startswith("model.")})
    ri = cases.rdt_inputs(cfg, 4, 20, seed=13, dtype=dtype)
    y = m(ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"])
    f32 = {k: v.float() for k, v in sd16.items()}
    rf = {k: (v.float() if v.is_floating_point() else v) for k, v in ri.items()}
    kw = dict(lang_mask=rf["lang_mask"], heads=cfg["heads"], horizon=cfg["horizon"], rms_mode="var")
    exact = orr.rdt_forward(f32, rf["x"], rf["freq"], rf["t"], rf["lang_c"], rf["img_c"], **kw)
    other = orr.rdt_forward(f32, rf["x"], rf["freq"], rf["t"], rf["lang_c"], rf["img_c"], **dict(kw, rms_mode="meansq"))
    scale = float(exact.abs().max())
    e = err(y, exact.numpy())
    assert err(other, exact.numpy()) > 10 * e or dname == "bf16"      # the two modes are distinguishable at this size
    if dname == "f32":
        assert e < 2e-4 * max(1.0, scale), e
    else:
        ref16 = orr.rdt_forward(sd16, ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"], lang_mask=ri["lang_mask"], heads=cfg["heads"],
                                horizon=cfg["horizon"], rms_mode="var")
        e_ref = err(ref16.float(), exact.numpy())
        print(f"[wide var] scale {scale:.2f}: |hip16-exact| {e:.3e}  |ref16-exact| {e_ref:.3e}")
        assert e <= max(1e-2 * scale, 1.5 * e_ref), (e, e_ref)


def test_engine_rejects_wrong_shapes():
    """The C driver indexes raw pointers with the packed config; the host side must refuse what the reference would refuse."""
    cfg = cases.RDT_TINY
    r = make_runner(cfg, torch.float32)
    ri = cases.rdt_inputs(cfg, 2, 12)
    a = [ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"]]
    bad_img = list(a); bad_img[2] = ri["img_tokens"][:, :-1]                 # one image token short
    bad_mask = list(a); bad_mask[1] = ri["lang_mask"][:, :-1]
    bad_freq = list(a); bad_freq[5] = ri["freq"][:1]
    for bad in (bad_img, bad_mask, bad_freq):
        with pytest.raises(ValueError):
            r.predict_action(*bad, x_init=ri["x_init"])
    with pytest.raises(ValueError):
        r.predict_action(*a, x_init=ri["x_init"][:, :-1])
    m = make_rdt(cfg, torch.float32)
    with pytest.raises(ValueError):
        m(ri["x"][:, :-1], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"])
    with pytest.raises(ValueError):
        m(ri["x"], ri["freq"], ri["t"], ri["lang_c"], ri["img_c"][:, 1:])


@pytest.mark.parametrize("B", [1, 4])
def test_cross_attention_fixed_maximum_matches_online_softmax(B):
    """The cached cross-attention with the load-time score bound (softmax against a FIXED maximum, vt_attn_kvt.hip) against its online form
    (vt_tune(6, 0)): same chunk within bf16 rounding of the attention output; B = 1 also takes the key-range-parts + combine path.  With the
    q / k norm gains scaled so that the bound exceeds 40 the launcher must fall back to the online form (bit-equal to it)."""
    from vlatouch import _lib as L
    cfg = cases.RDT_WIDE
    r = make_runner(cfg, torch.bfloat16, compute="bf16")        # bf16 probabilities: bounds up to 40 (fp16 ones: up to 10, tests/test_gpu_range_guard.py)
    ri = {k: v.to("cuda:0") for k, v in cases.rdt_inputs(cfg, B, 20, seed=5, dtype=torch.bfloat16).items()}
    args = (ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"])
    lib = L.lib()
    eng = r.engine()
    bounds = [8.0 * 1.02 * float(eng._weights[11 + 21 * i + 12].abs().max()) * float(eng._weights[11 + 21 * i + 13].abs().max()) for i in range(cfg["depth"])]
    assert 0 < max(bounds) <= 40, bounds
    try:
        fixed = r.predict_action(*args, x_init=ri["x_init"]).float()
        lib.vt_tune(6, 0)
        online = r.predict_action(*args, x_init=ri["x_init"]).float()
        lib.vt_tune(6, 1)
        scale = float(online.abs().max())
        assert err(fixed, online.cpu().numpy()) <= 1e-2 * scale, (err(fixed, online.cpu().numpy()), scale)
        # gains x 3 on both norms: bound x 9 > 40 -> online form regardless of the knob
        for i in range(cfg["depth"]):
            eng._weights[11 + 21 * i + 12].mul_(3.0)
            eng._weights[11 + 21 * i + 13].mul_(3.0)
        eng.repack()
        big_fixed = r.predict_action(*args, x_init=ri["x_init"]).float()
        lib.vt_tune(6, 0)
        big_online = r.predict_action(*args, x_init=ri["x_init"]).float()
        assert torch.equal(big_fixed, big_online)
    finally:
        lib.vt_tune(6, 1)
