"""GPU tests of the RANGE GUARD (round 6; VERDICT r5 "next" #1, ADVICE r5 medium 1 + 2).

The reference executes RDT in bf16 (models/rdt_runner.py:47-60,160; models/rdt/model.py:124).  The build's default computes a bf16 checkpoint with IEEE fp16
activations (3 more mantissa bits, 3 fewer exponent bits): `compute_dtype="auto"` keeps that only while the engine's sticky device word stays 0
(include/vlatouch.h, vt_rdt_set_range_flag; vlatouch.engine.RangeGuard / AutoRange) and otherwise re-runs in bf16.  Fixtures: the synthetic RDT weights of
tests/cases.py with a few OUTLIER channels, so that the fp32 oracle's activations exceed 65 504 where the fp16 engine stores 16 bits:
  "xn"  : attn.proj.bias = 3 000 in four channels, norm2.weight x 64 there -> the un-normalised hand-off operand x * gain = 192 000 (clamped: VT_RANGE_XN_SAT)
  "fc1" : ffn.fc1.bias = 2e5 in two channels -> GELU output 2e5 -> v_cvt_f16_f32 gives inf -> NaN through fc2 (VT_RANGE_NONFINITE at the solver update)
Both at hidden 2048 / batch 32 (M = 2 144 rows: the weights-in-registers tile with the fused RMSNorm hand-off, as in the benchmarked pipeline)."""
import warnings

import numpy as np
import pytest
import torch

from tests import cases
from tests.test_gpu_rdt import err, make_runner

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

CFG = cases.RDT_WIDE
B, LANG = 32, 20
ROWS = [0, 31]


def heavy_sd(kind):
    sd = {k: v.clone() for k, v in cases.rdt_sd(CFG, torch.float32).items()}
    if kind == "xn":
        ch = [5, 700, 1300, 2040]
        sd["model.blocks.0.attn.proj.bias"][ch] = 3000.0
        sd["model.blocks.0.norm2.weight"][ch] *= 64.0
    elif kind == "fc1":
        sd["model.blocks.1.ffn.fc1.bias"][[17, 1900]] = 2.0e5
    elif kind != "benign":
        raise ValueError(kind)
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}            # a bf16 checkpoint (values exactly representable in bf16)


def runner_with(sd, compute):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = make_runner(CFG, torch.bfloat16, compute=compute)
    r.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()})
    return r


def inputs():
    return {k: v.to("cuda:0") for k, v in cases.rdt_inputs(CFG, B, LANG, seed=31, dtype=torch.bfloat16).items()}


def run(r, ri, **kw):
    return r.predict_action(ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"], x_init=ri["x_init"],
                            return_fp32=True, **kw)


_ORACLE = {}


def oracle_rows(kind, sd):
    """fp32 oracle on rows ROWS of the batch (samples are independent), plus the same in bf16 arithmetic (the reference's own execution dtype)."""
    if kind in _ORACLE:
        return _ORACLE[kind]
    from oracle import rdt as orr
    ri = cases.rdt_inputs(CFG, B, LANG, seed=31, dtype=torch.bfloat16)
    sel = {k: (v[ROWS] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in ri.items()}
    f32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sel.items()}
    kw = dict(heads=CFG["heads"], horizon=CFG["horizon"], num_inference_steps=5)
    exact = orr.predict_action(sd, f32["lang_tokens"], f32["lang_mask"], f32["img_tokens"], f32["state_tokens"], f32["action_mask"], f32["freq"], f32["x_init"], **kw)
    sd16 = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    ref16 = orr.predict_action(sd16, sel["lang_tokens"], sel["lang_mask"], sel["img_tokens"], sel["state_tokens"], sel["action_mask"], sel["freq"], sel["x_init"], **kw)
    _ORACLE[kind] = (exact.float().numpy(), ref16.float().numpy())
    return _ORACLE[kind]


@pytest.mark.parametrize("kind,bit", [("xn", 1), ("fc1", 2)])
def test_heavy_tailed_weights_raise_the_flag_and_auto_falls_back_to_bf16(kind, bit):
    from vlatouch import _lib as L
    sd = heavy_sd(kind)
    ri = inputs()
    exact, ref16 = oracle_rows(kind, sd)
    scale = float(np.abs(exact).max())
    assert np.isfinite(exact).all() and scale > 0
    # pinned fp16: the guard records (sticky), the result is unusable (saturated or NaN) — exactly what "auto" must not hand out
    rf = runner_with(sd, "f16")
    out_f16 = run(rf, ri)
    bits = rf.overflowed()
    assert bits & bit, (kind, bits, L.range_names(bits))
    assert rf.overflowed() == bits                                       # sticky: not cleared by reading
    assert rf.overflowed(clear=True) == bits and rf.overflowed() == 0
    e_f16 = err(out_f16[ROWS], exact)
    print(f"[{kind}] pinned f16: flag {L.range_names(bits)}, |chunk - oracle| = {e_f16:.3e} (scale {scale:.2f})")
    del rf
    # pinned bf16 (the reference's dtype): unaffected — clean flag, within the reference's own bf16 error of the fp32 oracle
    rb = runner_with(sd, "bf16")
    out_bf = run(rb, ri)
    assert rb.overflowed() == 0
    e_bf, e_ref = err(out_bf[ROWS], exact), err(ref16, exact)
    print(f"[{kind}] pinned bf16: |chunk - oracle| = {e_bf:.3e}, oracle in bf16 arithmetic {e_ref:.3e}")
    assert torch.isfinite(out_bf).all()
    assert e_bf <= max(1e-2 * scale, 1.5 * e_ref), (kind, e_bf, e_ref, scale)
    # auto: starts in fp16, the first call trips the guard, warns, re-runs in bf16 and stays there
    ra = runner_with(sd, "auto")
    assert ra.compute_dtype == torch.float16
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        out_auto = run(ra, ri)
    assert ra.compute_dtype == torch.bfloat16 and ra.engine().dtype == torch.bfloat16
    assert torch.equal(out_auto, out_bf)                                  # the bf16 engine's result, bit for bit
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # no second warning, no second rebuild
        again = run(ra, ri)
    assert torch.equal(again, out_bf) and ra.overflowed() == 0


def test_benign_weights_keep_fp16_and_a_clean_flag():
    """The synthetic N(0, 1/sqrt(K)) weights of the benchmark: flag 0 after the (synchronously checked) first call and after more calls; "auto" == pinned "f16" bit for bit."""
    sd = heavy_sd("benign")
    ri = inputs()
    exact, _ = oracle_rows("benign", sd)
    ra, rf = runner_with(sd, "auto"), runner_with(sd, "f16")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a1 = run(ra, ri)
        a2 = run(ra, ri)
    f1 = run(rf, ri)
    assert ra.compute_dtype == torch.float16 and ra.overflowed() == 0 and rf.overflowed() == 0
    assert torch.equal(a1, f1) and torch.equal(a2, f1)
    scale = float(np.abs(exact).max())
    e = err(a1[ROWS], exact)
    print(f"[benign] auto (fp16): |chunk - oracle| = {e:.3e} (scale {scale:.2f})")
    assert e <= 1e-2 * max(1.0, scale)


def test_weights_outside_fp16_choose_bf16_at_load():
    sd = heavy_sd("benign")
    sd["model.blocks.0.ffn.fc2.weight"][3, 7] = 1.0e6                      # representable in bf16, inf in fp16
    with pytest.warns(RuntimeWarning, match="does not fit IEEE fp16"):
        r = make_runner(CFG, torch.bfloat16, compute="auto")
        r.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()})
        eng = r.engine()
    assert r.compute_dtype == torch.bfloat16 and eng.dtype == torch.bfloat16


def test_lag_one_readout_catches_an_overflow_after_the_first_call():
    """After a clean first call the guard is read without blocking (one call of lag): weights that change under the engine (here: in place, as the
    multi-GPU weight broadcast does) trip it on a later call; the call after that runs in bf16."""
    sd = heavy_sd("benign")
    ri = inputs()
    ra = runner_with(sd, "auto")
    run(ra, ri)
    assert ra.compute_dtype == torch.float16
    eng = ra.engine()
    fc1_b = eng._weights[11 + 21 * 1 + 18]                                 # block 1 ffn.fc1.bias (weight order: csrc/vt_rdt.hip)
    assert fc1_b.shape == (CFG["hidden"],) and fc1_b.dtype == torch.float32
    fc1_b[[17, 1900]] = 2.0e5
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        run(ra, ri)                                                      # overflows; its read-out is enqueued behind it
        torch.cuda.synchronize()
        run(ra, ri)                                                      # sees the completed read-out of the previous call -> falls back
    assert any("fp16 range" in str(x.message) for x in w), [str(x.message) for x in w]
    assert ra.compute_dtype == torch.bfloat16


@pytest.mark.parametrize("compute", ["bf16", "f16"])
def test_fully_masked_language_row_gives_zeros_and_a_flag_not_nan(compute):
    """a cross-attention row without any unmasked key has no softmax (torch's SDPA returns NaN there, blocks.py:116-123); the kernels write zeros
    for it and raise VT_RANGE_ATTN_EMPTY — the robot never receives NaN actions from an empty instruction."""
    from vlatouch import _lib as L
    sd = heavy_sd("benign")
    ri = inputs()
    ri["lang_mask"] = ri["lang_mask"].clone()
    ri["lang_mask"][3, :] = False
    r = runner_with(sd, compute)
    out = run(r, ri)
    assert torch.isfinite(out).all()
    assert r.overflowed() & L.RANGE_ATTN_EMPTY
    # the other samples are unaffected (bit-equal to a run without the empty row)
    r2 = runner_with(sd, compute)
    ri2 = inputs()
    ref = run(r2, ri2)
    keep = [i for i in range(B) if i != 3]
    assert torch.equal(out[keep], ref[keep])


def test_fp16_fixed_maximum_softmax_is_admitted_only_below_10():
    """P = 2^15 exp(s - B) must stay a normal fp16 number for every admissible score: the launcher takes the fixed-maximum form with
    fp16 probabilities only for bounds <= 10 (bf16: 40).  Synthetic q / k norm gains give bounds of 11 - 13 -> online form (bit-equal with the knob off);
    gains x 0.8 on both norms -> bounds <= 8.5 -> fixed form, same chunk within rounding."""
    from vlatouch import _lib as L
    sd = heavy_sd("benign")
    ri = inputs()
    lib = L.lib()
    r = runner_with(sd, "f16")
    eng = r.engine()
    assert 10.0 < min(eng.score_bounds) and max(eng.score_bounds) <= 16.0, eng.score_bounds      # (10.5, 11.3 with the synthetic gains 1 + 0.1 N(0, 1))
    try:
        a = run(r, ri)
        lib.vt_tune(6, 0)
        b = run(r, ri)
        lib.vt_tune(6, 1)
        assert torch.equal(a, b)                                         # both ran the online form
        for i in range(CFG["depth"]):
            eng._weights[11 + 21 * i + 12].mul_(0.8)
            eng._weights[11 + 21 * i + 13].mul_(0.8)
        eng.repack()
        assert max(eng.score_bounds) <= 10.0, eng.score_bounds
        fixed = run(r, ri)
        lib.vt_tune(6, 0)
        online = run(r, ri)
        scale = float(online.abs().max())
        assert not torch.equal(fixed, online)                            # different kernels
        assert err(fixed, online.cpu().numpy()) <= 2e-3 * scale, (err(fixed, online.cpu().numpy()), scale)
        assert r.overflowed() == 0
    finally:
        lib.vt_tune(6, 1)


def test_dinov2_giant_swiglu_saturation_is_flagged_and_falls_back(monkeypatch):
    """The gated FFN of dinov2-giant in the encoders' fp16 storage mode (csrc/vt_kernels.hip swiglu_kernel): silu(x1) * x2 beyond 65 504 is clamped AND
    recorded (VT_RANGE_GATE_SAT); DINOv2Encoder's default mode then rebuilds itself with bf16 storage and repeats the call."""
    from vlatouch import _lib as L, synth
    from vlatouch.engine import DinoEngine
    from residual_controller.visual_encoder import DINOv2Encoder
    D, layers, heads = 1536, 2, 24                                         # giant's width and FFN form, two blocks
    shapes = synth.dinov2_shapes(D, layers, swiglu=True)
    sd = {k: torch.from_numpy(v).clone() for k, v in synth.fill_state_dict(shapes, prefix="dinov2-giant.").items()}
    F = sd["encoder.layer.0.mlp.weights_out.weight"].shape[1]
    sd["encoder.layer.0.mlp.weights_in.bias"][[11, F + 11]] = 400.0         # x1 = x2 = 400 in one gate channel: silu(400) * 400 = 160 000
    imgs = torch.from_numpy(0.2 + 0.8 * synth.inputs_rng(3).random((2, 3, 224, 224), dtype=np.float32)).to("cuda:0")
    e16 = DinoEngine(sd, heads=heads, precision="fp16", device="cuda:0")
    y16 = e16.forward([imgs], nhwc=False)
    assert e16.overflowed() & L.RANGE_GATE_SAT
    ebf = DinoEngine(sd, heads=heads, precision="bf16", device="cuda:0")
    ybf = ebf.forward([imgs], nhwc=False)
    assert ebf.overflowed() == 0 and torch.isfinite(ybf).all()
    monkeypatch.delenv("VLATOUCH_DINO_PRECISION", raising=False)
    enc = DINOv2Encoder("facebook/dinov2-giant", device="cuda:0", precision="bf16", state_dict=sd)
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        y = enc.forward(imgs)
    assert enc.engine.adt == L.BF16 and torch.equal(y, ybf[0])
    del y16


def test_vit_towers_flag_a_non_finite_feature_and_fall_back():
    """The GELU ViT towers in their fp16 storage mode: an fc1 bias of 2e5 in one channel makes the GELU output inf in fp16 -> fc2 -> NaN in the fp32 residual stream -> the
    final norm's statistics are non-finite -> VT_RANGE_NONFINITE (csrc/vt_kernels.hip, rownorm kernels).  DINOv2Encoder / SiglipVisionTower in their default mode then
    rebuild themselves with bf16 storage and return the bf16 engine's (finite) result."""
    import types
    from vlatouch import _lib as L, synth
    from vlatouch.engine import DinoEngine
    from residual_controller.visual_encoder import DINOv2Encoder
    from models.multimodal_encoder.siglip_encoder import SiglipVisionTower
    # ---- DINOv2-small
    sd = {k: v.clone() for k, v in cases.dino_sd("small").items()}
    sd["encoder.layer.3.mlp.fc1.bias"][7] = 2.0e5
    imgs = torch.from_numpy(0.2 + 0.8 * synth.inputs_rng(5).random((2, 3, 224, 224), dtype=np.float32)).to("cuda:0")
    e16 = DinoEngine(sd, heads=6, precision="fp16", device="cuda:0")
    y16 = e16.forward([imgs], nhwc=False)
    assert e16.overflowed() & L.RANGE_NONFINITE and not torch.isfinite(y16).all()
    ebf = DinoEngine(sd, heads=6, precision="bf16", device="cuda:0")
    ybf = ebf.forward([imgs], nhwc=False)
    assert ebf.overflowed() == 0 and torch.isfinite(ybf).all()
    enc = DINOv2Encoder("facebook/dinov2-small", device="cuda:0", precision="bf16", state_dict=sd)
    assert enc.engine.adt == L.F16
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        y = enc.forward(imgs)
    assert enc.engine.adt == L.BF16 and torch.equal(y, ybf[0])
    # the benign weights keep fp16 and a clean word
    enc0 = DINOv2Encoder("facebook/dinov2-small", device="cuda:0", precision="bf16", state_dict=cases.dino_sd("small"))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        enc0.forward(imgs)
        enc0.forward(imgs)
    assert enc0.engine.adt == L.F16 and enc0.engine.overflowed() == 0
    # ---- SigLIP tower (out_all: the final norm runs on every token)
    c = synth.SIGLIP_CONFIGS["tiny"]
    cfg = dict(hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"], image_size=c["image_size"], patch_size=14)
    ssd = {k: v.clone() for k, v in cases.siglip_sd("tiny").items()}
    ssd["encoder.layers.0.mlp.fc1.bias"][3] = 2.0e5
    px = cases.siglip_pixels(2, c["image_size"], seed=8).to("cuda:0")
    args = types.SimpleNamespace(mm_vision_select_feature="patch")
    tower = SiglipVisionTower("synthetic", args, device="cuda:0", precision="bf16", state_dict=ssd, config=cfg)
    assert tower.engine.adt == L.F16
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        out = tower(px)
    assert tower.engine.adt == L.BF16 and torch.isfinite(out.float()).all()
