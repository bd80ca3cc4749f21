#!/bin/bash
# A/B of the online softmax's row maximum in the cached cross-attention: v_permlane swaps (product) vs __shfl_xor (-DVLATOUCH_KVT_MAX_SHFL), same box, interleaved
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_KVT_MAX_SHFL -c vt_attn_kvt.hip -o /tmp/vt_attn_kvt_shfl.o 2>/dev/null
O=$(ls build/*.o | grep -v vt_attn_kvt.o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvlatouch_shfl.so $O /tmp/vt_attn_kvt_shfl.o
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rdt.py -q -m gpu -k "cross_attention or wide or predict_action" 2>&1 | tail -1
for r in 1 2; do for lib in "" /tmp/libvlatouch_shfl.so; do
  echo "== online softmax (rms var), lib=${lib:-product(permlane)}"
  VLATOUCH_LIB=$lib python bench.py --no-cpu-baseline --alt-compute-steps 0 --rms-mode var 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=[x for x in d['roofline_other'] if 'attn_kvt' in x['kernel']][0]; print(d['value'], d['latency_mode']['chunks_per_s'], 'kvt avg us', r['avg_launch_us'])"
done; done
