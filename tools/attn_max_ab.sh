#!/bin/bash
# A/B of the ViT / RDT self-attention's cross-lane reductions: permlane swaps (product) vs __shfl_xor (-DVLATOUCH_ATTN_SHFL), same box
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_ATTN_SHFL -c vt_attn.hip -o /tmp/vt_attn_shfl.o 2>/dev/null
O=$(ls build/*.o | grep -v "vt_attn.o"); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvlatouch_ashfl.so $O /tmp/vt_attn_shfl.o
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_primitives.py -q -m gpu -k "attention" 2>&1 | tail -1
for r in 1 2; do for lib in "" /tmp/libvlatouch_ashfl.so; do
  echo "== lib=${lib:-product(permlane)}"
  VLATOUCH_LIB=$lib python tools/attn_bench.py 2>&1 | grep "TF/s"
  for w in siglip dino_mlp; do VLATOUCH_LIB=$lib python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'])"; done
done; done
