"""CPU-only checks: the C-ABI library loads and exports every symbol include/vlatouch.h declares (no compute calls),
host-side logic (DPM-Solver++ coefficients, parameter containers, EMA semantics, input layout handling, sharding)
and the world_size-2 gloo path of the weight broadcast."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import cases

ROOT = cases.ROOT
HEADER = os.path.join(ROOT, "include", "vlatouch.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vlatouch import _lib
    lib = _lib.lib()                     # binds every name in SIGNATURES, raises if one is missing
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    unbound = [s for s in syms if s not in _lib.SIGNATURES and s not in ("vt_last_error", "vt_version")]
    assert not unbound, f"declared in the header but without a ctypes signature: {unbound}"
    assert lib.vt_version() >= 100
    assert lib.vt_last_error() is not None


def test_lzf_library_exports_its_header():
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "vtlzf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    syms = sorted(set(re.findall(r"\b(vt_lzf_[a-z0-9_]+)\s*\(", hdr)))
    assert syms == ["vt_lzf_compress", "vt_lzf_decompress"]
    lib = C.CDLL(os.path.join(ROOT, "vla-touch_amd", "vlatouch", "libvtlzf.so"))
    assert all(hasattr(lib, s) for s in syms)


def test_ctypes_structs_match_c_layout():
    """sizeof/offsets of the parameter blocks, computed by compiling a tiny C++ probe against the real headers."""
    import ctypes as C
    from vlatouch import _lib
    probe = r'''
#include <cstdio>
#include <cstddef>
#include <hip/hip_runtime.h>
#include "vt_kernels.h"
#include "../../include/vlatouch.h"
int main() {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(VtGemmParams), offsetof(VtGemmParams, cmap_T), sizeof(VtGnParams), sizeof(VtAttnParams),
         sizeof(vt_unet_desc), sizeof(vt_dino_desc), sizeof(vt_rdt_desc));
  return 0;
}'''
    d = os.path.join(ROOT, "vla-touch_amd", "csrc")
    src = os.path.join(d, "build", "_probe.cpp")
    os.makedirs(os.path.dirname(src), exist_ok=True)
    open(src, "w").write(probe)
    exe = src.replace(".cpp", "")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-I", d, src, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("hipcc host probe did not build: " + r.stderr[-300:])
    out = subprocess.run([exe], capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    want = [C.sizeof(_lib.GemmParams), _lib.GemmParams.cmap_T.offset, C.sizeof(_lib.GnParams), C.sizeof(_lib.AttnParams), C.sizeof(_lib.UnetDesc),
            C.sizeof(_lib.DinoDesc), C.sizeof(_lib.RdtDesc)]
    assert got == want, (got, want)


def test_product_refuses_cpu_devices():
    from vlatouch import _lib
    from residual_controller.bridge_controller import DiffusionController
    with pytest.raises(_lib.VtError):
        _lib.require_gpu("cpu")
    with pytest.raises((_lib.VtError, FileNotFoundError)):
        DiffusionController(device="cpu")


def test_dpm_coefficients_match_oracle_scheduler():
    from oracle import dpm_solver
    from vlatouch import dpm
    for n in (3, 5, 10):
        ts, coef = dpm.schedule(1000, "squaredcos_cap_v2", n)
        s = dpm_solver.DPMSolverPP2M(1000, "squaredcos_cap_v2", "sample")
        s.set_timesteps(n)
        assert ts == s.timesteps
        ref = s.coefficients()
        for i in range(n):
            for j, k in enumerate(("a", "b0", "b1", "alpha_s", "sigma_s")):
                assert abs(coef[i, j] - ref[i][k]) < 1e-5 * max(1.0, abs(ref[i][k])), (n, i, k, coef[i, j], ref[i][k])
        # folded update == the scheduler's step on random data
        g = torch.Generator().manual_seed(n)
        x = torch.randn(4, 7, generator=g)
        prev = None
        for i in range(n):
            m = torch.randn(4, 7, generator=g)
            want = s.step(m, x)
            got = coef[i, 0] * x + coef[i, 1] * m + (coef[i, 2] * prev if prev is not None else 0)
            assert float((got - want).abs().max()) < 1e-5
            x, prev = want, m
    with pytest.raises(NotImplementedError):
        dpm.schedule(1000, "nope", 5)


def test_param_module_and_ema_semantics():
    from vlatouch.module import ExponentialMovingAverage, ParamModule
    m = ParamModule({"0.weight": (4, 3), "0.bias": (4,)})
    v0 = m.version
    with pytest.raises(RuntimeError):
        m.load_state_dict({"0.weight": torch.zeros(4, 3)})
    with pytest.raises(RuntimeError):
        m.load_state_dict({"0.weight": torch.zeros(5, 3), "0.bias": torch.zeros(4)})
    m.load_state_dict({"0.weight": torch.ones(4, 3), "0.bias": torch.zeros(4)})
    assert m.version > v0 and list(m.state_dict().keys()) == ["0.weight", "0.bias"]
    ema = ExponentialMovingAverage(m.parameters(), decay=0.75).bind(m)
    ema.load_state_dict({"decay": 0.75, "num_updates": 3, "shadow_params": [torch.full((4, 3), 2.0), torch.full((4,), 5.0)], "collected_params": None})
    with ema.average_parameters():
        assert float(m.state_dict()["0.weight"][0, 0]) == 2.0 and float(m.state_dict()["0.bias"][0]) == 5.0
    assert float(m.state_dict()["0.weight"][0, 0]) == 1.0
    assert sorted(ema.state_dict().keys()) == ["collected_params", "decay", "num_updates", "shadow_params"]
    with pytest.raises(ValueError):
        ema.load_state_dict({"decay": 0.75, "num_updates": 0, "shadow_params": [torch.zeros(1)]})
    lazy = ParamModule({"w": (2, 2)}, materialize=False)
    big = torch.ones(2, 2, dtype=torch.bfloat16)
    lazy.load_state_dict({"w": big}, assign=True)
    assert lazy.state_dict()["w"] is big or lazy.state_dict()["w"].data_ptr() == big.data_ptr()


def test_dino_input_layout_rules():
    from residual_controller.visual_encoder import DINOv2Encoder
    lay = DINOv2Encoder._layout
    t, nhwc, ps = lay(torch.zeros(2, 3, 28, 28))
    assert (tuple(t.shape), nhwc, ps) == ((2, 3, 28, 28), False, 1.0)
    t, nhwc, ps = lay(torch.zeros(2, 28, 28, 3))
    assert nhwc and ps == 1.0
    t, nhwc, ps = lay(torch.zeros(2, 4, 28, 28, 3))
    assert tuple(t.shape) == (8, 28, 28, 3) and nhwc
    t, nhwc, ps = lay(np.zeros((2, 28, 28, 3), dtype=np.uint8))
    assert t.dtype == torch.uint8 and nhwc and abs(ps - 1 / 255.0) < 1e-12
    with pytest.raises(ValueError):
        lay(torch.zeros(3, 28, 28))


def test_shard_range_partitions_episodes():
    from vlatouch.dist import shard_range
    for n, w in ((256, 8), (10, 4), (3, 8), (33, 2)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import torch, torch.distributed as dist
from vlatouch.dist import broadcast_tensors, gather_results, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = torch.Generator().manual_seed(100 + rank)
ts = [torch.randn(5, 7, generator=g), torch.randn(3, generator=g), torch.randn(2, 2, generator=g).to(torch.bfloat16), None,
      torch.randn(6, 4, generator=g)[:, ::2]]
ref = torch.Generator().manual_seed(100)
want = [torch.randn(5, 7, generator=ref), torch.randn(3, generator=ref), torch.randn(2, 2, generator=ref).to(torch.bfloat16), None,
        torch.randn(6, 4, generator=ref)[:, ::2]]
n = broadcast_tensors(ts, src=0, bucket_bytes=64)
assert n == sum(t.numel() * t.element_size() for t in ts if t is not None)
for a, b in zip(ts, want):
    assert (a is None and b is None) or torch.equal(a, b), (rank, a, b)
lo, hi = shard_range(10, rank, world)
local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1).repeat(1, 2, 3)
outs = gather_results(local, dst=0)
if rank == 0:
    assert torch.equal(torch.cat(outs)[:, 0, 0], torch.arange(10.0))
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
'''


def test_weight_broadcast_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "pkg": cases.PKG})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_rms_mode_is_an_explicit_choice(monkeypatch):
    """timm's RmsNorm arithmetic is third-party and unpinned (models/rdt/blocks.py:22): argument > config > environment > warned default."""
    import warnings
    from models import rdt_runner as rr
    monkeypatch.delenv("VLATOUCH_TIMM_RMSNORM", raising=False)
    assert rr.resolve_rms_mode("var", {}) == "var"
    assert rr.resolve_rms_mode(None, {"rms_norm": "var"}) == "var"
    assert rr.resolve_rms_mode("meansq", {"rms_norm": "var"}) == "meansq"
    monkeypatch.setenv("VLATOUCH_TIMM_RMSNORM", "var")
    assert rr.resolve_rms_mode(None, {}) == "var"
    monkeypatch.delenv("VLATOUCH_TIMM_RMSNORM")
    rr._warned_default_rms = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert rr.resolve_rms_mode(None, {}) == "meansq"
    assert any("timm==1.0.3" in str(x.message) for x in w)
    with pytest.raises(ValueError):
        rr.resolve_rms_mode("l2", {})


def test_fused_unet_plan_coverage_is_decided_per_shape_on_the_host():
    """vt_unet_fused_plan_bytes dry-runs the fused sampler plan (tile choice per convolution, final-kernel LDS budget) without a GPU: the
    controller's U-Net (256, 512, 512) is covered at the reference's two cadences (16- and 48-tick chunks, bridge_controller.py /
    scripts/franka_inference_eef.py) and at T = 8 .. 64; shapes the plan cannot run report 0 and stay on the launch-per-op driver."""
    import ctypes as C
    from vlatouch import _lib as L
    lib = L.lib()

    def handle(dims, cdt=2, adt=0):
        d = L.UnetDesc()
        d.nets, d.input_dim, d.input_pad, d.cond_dim, d.dsed, d.n_groups, d.ksize, d.n_levels = 2, 10, 32, 256, 256, 8, 5, len(dims)
        for i, v in enumerate(dims):
            d.dims[i] = v
        d.cdt, d.adt = cdt, adt                         # VT_F32X3 weights, VT_F32 activations: the split-bf16 mode the fused path exists for
        n = lib.vt_unet_num_weights(C.byref(d))
        fake = (C.c_void_p * n)(*([0x1000] * n))        # never dereferenced: plan sizing is host arithmetic
        h = C.c_void_p()
        L.check(lib.vt_unet_create(C.byref(d), fake, n, C.byref(h)), "vt_unet_create")
        return h
    h = handle((256, 512, 512))
    for B in (1, 5, 32):
        for T in (8, 12, 16, 24, 32, 48, 64):
            assert lib.vt_unet_fused_plan_bytes(h, B, T, 10) > 0, (B, T)
        for T in (10, 18, 50, 68):                      # not a multiple of 4, or past 64 ticks
            assert lib.vt_unet_fused_plan_bytes(h, B, T, 10) == 0, (B, T)
    assert lib.vt_unet_workspace_bytes(h, 32, 48) >= lib.vt_unet_fused_plan_bytes(h, 32, 48, 64) > 0      # sized before any vt_unet_fused_pack
    wide = handle((512, 512))
    assert lib.vt_unet_fused_plan_bytes(wide, 3, 16, 10) > 0
    assert lib.vt_unet_fused_plan_bytes(wide, 3, 32, 10) == 0       # final kernel: 2 * (32 + 10) * 512 * 4 B = 172 KB of LDS
    narrow = handle((64, 64, 64, 64))
    assert lib.vt_unet_fused_plan_bytes(narrow, 64, 8, 10) > 0      # tile choice bounded to 256 GroupNorm units per block
    fp32 = handle((256, 512, 512), cdt=0)
    assert lib.vt_unet_fused_plan_bytes(fp32, 4, 16, 10) == 0       # exact-fp32 mode has no fused path


def test_interpolant_schedule_tables_match_the_oracle():
    """The mirror's epsilon / gamma / gamma_der / gamma_inv are string-keyed table lookups (bridge_model.py:59-101 of the reference as data);
    every key against the oracle's restatement (itself pinned to the reference through g2's trajectories), and unknown keys raise NotImplementedError."""
    import pytest
    from oracle import interpolant as oi
    from residual_controller.bridge import bridge_model as bm
    t = torch.linspace(0.001, 0.999, 97)
    si = bm.StochasticInterpolants.__new__(bm.StochasticInterpolants)
    si.gamma_inv_max = 200.0
    assert set(bm._GAMMA_OF_T) == set(bm._GAMMA) and set(bm._EPSILON_OF_T) == set(bm._EPS)
    for k in bm._EPS:
        si.epsilon_type = k
        assert torch.allclose(si.epsilon(t), oi.epsilon(t, k), rtol=0, atol=1e-6), k
    for k in bm._GAMMA:
        si.gamma_type = k
        for name in ("gamma", "gamma_der", "gamma_inv"):
            got, ref = getattr(si, name)(t), getattr(oi, name)(t, k)
            assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6), (k, name, float((got - ref).abs().max()))
    si.gamma_type = si.epsilon_type = "nope"
    for name in ("gamma", "gamma_der", "gamma_inv", "epsilon"):
        with pytest.raises(NotImplementedError):
            getattr(si, name)(t)


def test_auto_range_policy_first_call_synchronous_then_lagged(monkeypatch):
    """vlatouch.engine.AutoRange (the compute_dtype="auto" policy, round 6) on a stand-in engine: the first call reads the guard synchronously, later calls through the
    non-blocking read-out, nothing is read while a hipGraph is being captured, and a non-fp16 engine is never consulted."""
    import warnings
    from vlatouch import _lib
    from vlatouch.engine import AutoRange

    class Eng:
        def __init__(self):
            self.sync, self.poll, self.bits = 0, 0, 0

        def overflowed(self, clear=False):
            self.sync += 1
            return self.bits

        def range_poll(self):
            self.poll += 1
            return self.bits

    capturing = {"on": False}
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing["on"])
    e, rg = Eng(), AutoRange("unit")
    assert rg.after(e, is_f16=False) == 0 and (e.sync, e.poll) == (0, 0) and not rg.checked       # bf16 / fp32 engines: nothing to guard
    capturing["on"] = True
    assert rg.after(e, True) == 0 and (e.sync, e.poll) == (0, 0) and not rg.checked               # capture: no read, and the first real call still gets its check
    capturing["on"] = False
    assert rg.after(e, True) == 0 and (e.sync, e.poll) == (1, 0) and rg.checked
    assert rg.after(e, True) == 0 and (e.sync, e.poll) == (1, 1)
    e.bits = _lib.RANGE_XN_SAT | _lib.RANGE_NONFINITE
    bits = rg.after(e, True)
    assert bits == 3 and (e.sync, e.poll) == (1, 2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        rg.fall_back(bits)
    assert rg.fell_back and rg.bits == 3 and len(w) == 1 and "xn_saturated, nonfinite" in str(w[0].message) and issubclass(w[0].category, RuntimeWarning)
    assert _lib.range_names(8 | 4) == ["gate_saturated", "attention_row_empty"]


def test_runner_compute_dtype_strings():
    """RDTRunner(compute_dtype=...) accepts "auto" (default) / "f16" / "bf16" for a bf16 model, computes fp32 models in fp32, and rejects anything else (no GPU needed:
    the engine is built lazily)."""
    from models.rdt_runner import RDTRunner
    cfg = {"rdt": {"hidden_size": 256, "depth": 1, "num_heads": 4, "rms_norm": "meansq"}, "lang_adaptor": "linear", "img_adaptor": "linear", "state_adaptor": "linear",
           "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": 5, "beta_schedule": "squaredcos_cap_v2", "prediction_type": "sample"}}
    kw = dict(action_dim=128, pred_horizon=8, config=cfg, lang_token_dim=32, img_token_dim=32, state_token_dim=128, max_lang_cond_len=8, img_cond_len=8, init_weights=False)
    r = RDTRunner(dtype=torch.bfloat16, **kw)
    assert r.compute_dtype == torch.float16 and r._range is not None
    assert RDTRunner(dtype=torch.bfloat16, compute_dtype="f16", **kw)._range is None
    rb = RDTRunner(dtype=torch.bfloat16, compute_dtype="bf16", **kw)
    assert rb.compute_dtype == torch.bfloat16 and rb._range is None
    rf = RDTRunner(dtype=torch.float32, **kw)
    assert rf.compute_dtype == torch.float32 and rf._range is None
    with pytest.raises(ValueError):
        RDTRunner(dtype=torch.bfloat16, compute_dtype="fp8", **kw)
