#!/usr/bin/env python
"""Phase times inside the fused U-Net convolution (csrc/vt_uconv.hip) launch by launch: every block stamps s_memrealtime (100 MHz) at
start / gathers done / barrier 1 / statistics done / operands built / k-loop done / stores issued (vt_uconv_set_timing).
    python tools/uconv_phases.py [B]      -> per launch: blocks, span of the launch, median per-phase microseconds"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'vla-touch_amd')
from tests import cases
from vlatouch.engine import UNetEngine
from vlatouch import _lib as L
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ema = cases.si_net_sd("ema")
sds = [{k[len(n) + 1:]: v for k, v in ema.items() if k.startswith(n + ".")} for n in ("v_net", "s_net")]
eng = UNetEngine(sds, precision="bf16", device=dev)
x, cond = cases.unet_inputs(B, 16, seed=5)
for _ in range(3): eng.forward(x, 0.37, cond)
NL = 32
buf = torch.zeros(NL * 2048 * 8, dtype=torch.int64, device=dev)
lib = ctypes.CDLL(L.LIB_PATH)
lib.vt_uconv_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
torch.cuda.synchronize()
lib.vt_uconv_set_timing(buf.data_ptr(), NL)
eng.forward(x, 0.37, cond)
torch.cuda.synchronize()
lib.vt_uconv_set_timing(None, 0)
t = buf.cpu().numpy().reshape(NL, 2048, 8)
names = ["gather", "bar1", "stats", "build", "kloop", "store"]
print(f"B={B}: per launch  blocks  span_us | median us per phase: " + " ".join(names) + " | total")
for l in range(NL):
    live = t[l, :, 0] != 0
    if not live.any(): continue
    tt = t[l, live].astype(np.float64)
    span = (tt[:, 6].max() - tt[:, 0].min()) / 100.0
    ph = np.median(np.diff(tt[:, :7], axis=1), axis=0) / 100.0
    J = int(tt[0, 7]) & 255
    start_spread = (tt[:, 0].max() - tt[:, 0].min()) / 100.0
    print(f"  {l:2d} J={J} blocks={int(live.sum()):5d} span={span:6.2f} start_spread={start_spread:5.2f} | " + " ".join(f"{v:5.2f}" for v in ph) + f" | {ph.sum():5.2f}")
