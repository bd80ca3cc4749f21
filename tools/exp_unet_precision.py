import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import numpy as np, torch
from tests import cases
from vlatouch.engine import UNetEngine
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
def split_nets(sd, nets=("v_net", "s_net")):
    return [{k[len(n) + 1:]: v for k, v in sd.items() if k.startswith(n + ".")} for n in nets]
ema = cases.si_net_sd("ema")
for T in (16, 32):
    g = np.load(f"{cases.GOLDEN}/g2_si_traj_T{T}.npz")
    x0, cond, _ = cases.si_inputs(2, T)
    for prec, adt in (("fp32", None), ("bf16", None), ("bf16_plain", None)):
        eng = UNetEngine(split_nets(ema), precision=prec, act_dtype=adt, device=dev)
        xT, traj = eng.sample(x0, cond, torch.from_numpy(g["z"]), 10, 0.03, record=True)
        e = np.abs(traj.cpu().numpy() - g["traj"]).reshape(11, -1).max(1)
        # single forward error
        out = eng.forward(x0, 0.5, cond).cpu().numpy()
        print(f"T={T} prec={prec} adt={adt}: traj err per step {np.array2string(e, precision=4)}", flush=True)
