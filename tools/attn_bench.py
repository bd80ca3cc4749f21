"""ViT self-attention (attn16_kernel) at the SigLIP-so400m and DINOv2-B shapes of the bench: time per layer and MFMA rate; with a bench build, the
timing-only ablations of tools/attn_abl.sh (VLATOUCH_ATTN_ABL)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
import torch
from vlatouch import ops
dev = torch.device("cuda:0")


def run(name, B, N, H, hd, hd_real, dt):
    qkv = torch.randn(B, N, 3, H, hd, device=dev).to(dt)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    f = lambda: ops.attention(q, k, v, scale=hd_real ** -0.5)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 4.0 * B * H * N * N * hd_real
    print(f"{name:28s} B={B} N={N} H={H} hd={hd}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s (useful flops)", flush=True)


run("siglip so400m (6 x 32 imgs)", 192, 729, 16, 80, 72, torch.float16)
run("dinov2-b (2 x 32 imgs)", 64, 257, 12, 64, 64, torch.float16)
run("rdt self-attention (B = 32)", 32, 67, 32, 64, 64, torch.bfloat16)
