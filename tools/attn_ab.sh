#!/bin/bash
# A/B of the two 16-bit self-attention kernels (VLATOUCH_ATTN16=1: attn16_kernel, 2: attn16u_kernel): parity tests, isolated time, siglip / dino_mlp lines
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "attention or attn" 2>&1 | tail -3
for a in 1 2 1 2; do echo "== VLATOUCH_ATTN16=$a"; VLATOUCH_ATTN16=$a python tools/attn_bench.py 2>&1 | grep "TF/s"; done
for a in 1 2 1 2; do for w in siglip dino_mlp; do echo "== VLATOUCH_ATTN16=$a $w"; VLATOUCH_ATTN16=$a python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; done; done
