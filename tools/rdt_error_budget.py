"""Where does the 16-bit RDT path's error against the fp32 oracle come from?  A CPU study (no GPU, no product code): the oracle's arithmetic
(oracle/rdt.py, restated here with a rounding hook at every point where the HIP path stores a 16-bit activation) run on ONE episode of the RDT-1B
shape, once in exact fp32 and once per rounding class with ONLY that class rounded to bf16 (or IEEE fp16) — max |x0 - exact| of the 5-step
DPM-Solver++ result per class.  Weights: bf16-rounded random-init values in every run (the reference's dtype), fp32 accumulation everywhere.

    python tools/rdt_error_budget.py [--depth 28] [--img-len 4374] [--steps 5]

Rounding classes = the HIP path's 16-bit stores (csrc/vt_rdt.hip): xn (RMSNorm outputs = Linear operands), qkv (projected + head-normed q / k / v of the
self-attention, cross q), kv (cached condition K / V), p (softmax probabilities fed to P V), att (attention outputs), hid (GELU(fc1)), cond (adapted
condition tokens + position embeddings), emb (timestep / frequency embeddings, state and action tokens), out (final-layer hidden + projection), state
(solver state between steps)."""
import argparse, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
import torch.nn.functional as F
from oracle import dpm_solver
from oracle.rdt import rms_norm, timestep_embedding
from vlatouch import synth

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=28)
ap.add_argument("--img-len", type=int, default=4374)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--classes", default="xn,qkv,kv,p,att,hid,cond,emb,out,state,all")
ap.add_argument("--types", default="bf16,f16")
args = ap.parse_args()
if args.threads:
    torch.set_num_threads(args.threads)
torch.set_grad_enabled(False)

D, H, HOR, ADIM, LANG, IMG = 2048, 32, 64, 128, 4096, 1152
cfg = dict(hidden=D, depth=args.depth, heads=H, horizon=HOR, action_dim=ADIM, lang_token_dim=LANG, img_token_dim=IMG, state_token_dim=128,
           max_lang_cond_len=1024, img_cond_len=args.img_len)
sd = {k: v.float() for k, v in synth.fill_state_dict_device(synth.rdt_runner_shapes(**cfg), "cpu", torch.bfloat16, seed=7).items()}
g = torch.Generator().manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
lang, img, state, x0 = rn(1, 32, LANG), rn(1, args.img_len, IMG), rn(1, 1, 128), rn(1, HOR, ADIM)
amask = torch.zeros(1, 1, 128); amask[:, :, :10] = 1.0
freq = torch.full((1,), 10.0)

ACTIVE, RT = set(), torch.bfloat16
def Q(cls, x):
    return x.to(RT).float() if (cls in ACTIVE or "all" in ACTIVE) else x

def lin(x, p): return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])
def embed(p, t):
    e = Q("emb", timestep_embedding(t, 256, torch.float32))
    return Q("emb", lin(Q("emb", F.silu(lin(e, p + ".mlp.0"))), p + ".mlp.2"))
def sdpa(q, k, v):
    s = (q @ k.transpose(-1, -2)) * 0.125
    p = torch.softmax(s, dim=-1)
    # the HIP kernels normalise AFTER P V with the fp32 row sum; P itself (un-normalised exp) is what gets rounded: same relative error
    return Q("p", p) @ v
def adaptor(p, x):
    i = 0
    while f"{p}.{i}.weight" in sd:
        if i > 0: x = Q("cond" if p != "state_adaptor" else "emb", F.gelu(x, approximate="tanh"))
        x = lin(x, f"{p}.{i}")
        i += 2
    return x

def run():
    lang_c = Q("cond", Q("cond", adaptor("lang_adaptor", lang)) + sd["model.lang_cond_pos_embed"][:, :32])
    img_c = Q("cond", Q("cond", adaptor("img_adaptor", img)) + sd["model.img_cond_pos_embed"])
    st_tok = Q("emb", adaptor("state_adaptor", torch.cat([state, amask], dim=2)))
    fe = embed("model.freq_embedder", freq).unsqueeze(1)
    # cached condition K / V per block (constant over the steps)
    kvs = []
    for i in range(args.depth):
        b = f"model.blocks.{i}.cross_attn"
        c = lang_c if i % 2 == 0 else img_c
        kv = lin(c, b + ".kv").reshape(1, c.shape[1], 2, H, 64).permute(2, 0, 3, 1, 4)
        kvs.append((Q("kv", rms_norm(kv[0], sd[b + ".k_norm.weight"])), Q("kv", kv[1])))
    sched = dpm_solver.DPMSolverPP2M(1000, "squaredcos_cap_v2", "sample")
    sched.set_timesteps(args.steps)
    noisy = x0.clone()
    am = amask.expand(-1, HOR, -1)
    for t in sched.timesteps:
        a = Q("emb", adaptor("state_adaptor", Q("emb", torch.cat([noisy, am], dim=2))))
        te = embed("model.t_embedder", torch.tensor([int(t)])).unsqueeze(1)
        x = torch.cat([te, fe, st_tok, a], dim=1) + sd["model.x_pos_embed"]
        for i in range(args.depth):
            b = f"model.blocks.{i}"
            xn = Q("xn", rms_norm(x, sd[b + ".norm1.weight"]))
            qkv = lin(xn, b + ".attn.qkv").reshape(1, -1, 3, H, 64).permute(2, 0, 3, 1, 4)
            q = Q("qkv", rms_norm(qkv[0], sd[b + ".attn.q_norm.weight"])); k = Q("qkv", rms_norm(qkv[1], sd[b + ".attn.k_norm.weight"])); v = Q("qkv", qkv[2])
            o = Q("att", sdpa(q, k, v).transpose(1, 2).reshape(1, -1, D))
            x = lin(o, b + ".attn.proj") + x
            xn = Q("xn", rms_norm(x, sd[b + ".norm2.weight"]))
            q = Q("qkv", rms_norm(lin(xn, b + ".cross_attn.q").reshape(1, -1, H, 64).permute(0, 2, 1, 3), sd[b + ".cross_attn.q_norm.weight"]))
            o = Q("att", sdpa(q, kvs[i][0], kvs[i][1]).permute(0, 2, 1, 3).reshape(1, -1, D))
            x = lin(o, b + ".cross_attn.proj") + x
            xn = Q("xn", rms_norm(x, sd[b + ".norm3.weight"]))
            x = lin(Q("hid", F.gelu(lin(xn, b + ".ffn.fc1"), approximate="tanh")), b + ".ffn.fc2") + x
        xn = Q("out", rms_norm(x, sd["model.final_layer.norm_final.weight"]))
        out = Q("out", lin(Q("out", F.gelu(lin(xn, "model.final_layer.ffn_final.fc1"), approximate="tanh")), "model.final_layer.ffn_final.fc2"))[:, -HOR:]
        noisy = Q("state", sched.step(out, noisy))
    return noisy * am

t0 = time.time()
exact = run()
print(f"# RDT D={D} depth={args.depth} img_len={args.img_len} steps={args.steps}: exact fp32 run {time.time() - t0:.1f} s, output scale {float(exact.abs().max()):.3f}", flush=True)
for tname in args.types.split(","):
    RT = {"bf16": torch.bfloat16, "f16": torch.float16}[tname]
    for cls in args.classes.split(","):
        ACTIVE = {cls}
        got = run()
        e = float((got - exact).abs().max())
        print(f"{tname:5s} only {cls:6s} rounded: max|x0 - exact| = {e:.3e}  ({e / float(exact.abs().max()):.2e} of scale)", flush=True)
