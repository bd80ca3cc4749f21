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
