"""Phase times inside attn16u_kernel (bench build: `make DEFS=-DVLATOUCH_BENCH_BUILD`; tools/attn_phases.sh builds it on the box): every block stamps s_memrealtime
(100 MHz) at entry / first data landed (Q + first two K, V stages) / key loop done / stores issued.  Per shape: launch span, spread of the block start times, median phases."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import numpy as np, torch
from vlatouch import ops, _lib as L
dev = torch.device("cuda:0")
lib = ctypes.CDLL(L.LIB_PATH)
lib.vt_attn_set_timing.argtypes = [ctypes.c_void_p]


def run(name, B, N, H, hd, hd_real, dt):
    qkv = torch.randn(B, N, 3, H, hd, device=dev).to(dt)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    f = lambda: ops.attention(q, k, v, scale=hd_real ** -0.5)
    for _ in range(3):
        f()
    buf = torch.zeros(4096 * 4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib.vt_attn_set_timing(buf.data_ptr())
    f()
    torch.cuda.synchronize()
    lib.vt_attn_set_timing(None)
    t = buf.cpu().numpy().reshape(4096, 4).astype(np.float64)
    t = t[t[:, 0] != 0]
    span = (t[:, 3].max() - t[:, 0].min()) / 100.0
    st = (t[:, 0] - t[:, 0].min()) / 100.0
    ph = np.diff(t, axis=1) / 100.0
    print(f"{name}: blocks stamped {len(t)} (of {B * H * ((N + 127) // 128)}), span {span:.2f} us; block start after launch: median {np.median(st):.2f} p90 {np.percentile(st, 90):.2f} max {st.max():.2f} us;"
          f" per block median (p90) us: first data {np.median(ph[:, 0]):.2f} ({np.percentile(ph[:, 0], 90):.2f}), key loop {np.median(ph[:, 1]):.2f} ({np.percentile(ph[:, 1], 90):.2f}),"
          f" finish {np.median(ph[:, 2]):.2f} ({np.percentile(ph[:, 2], 90):.2f}); block life {np.median(t[:, 3] - t[:, 0]) / 100:.2f}", flush=True)


run("rdt self-attention (B = 32)", 32, 67, 32, 64, 64, torch.bfloat16)
run("dinov2-b (2 x 32 imgs)", 64, 257, 12, 64, 64, torch.float16)
run("siglip so400m (1 x 32 imgs)", 32, 729, 16, 80, 72, torch.float16)
