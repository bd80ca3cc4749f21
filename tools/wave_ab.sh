#!/bin/bash
# A/B of the wave-wide reductions (vt_common.h wave_sum / wave_max): DPP + permlane swaps (product) vs the __shfl_xor butterfly (-DVLATOUCH_WAVE_SHFL), same box
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
rm -rf /tmp/build_wshfl; mkdir -p /tmp/build_wshfl
for f in *.hip; do e=""; [ $f = vt_uconv.hip ] && e="-Xclang -target-feature -Xclang -packed-fp32-ops"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_WAVE_SHFL $e -c $f -o /tmp/build_wshfl/${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvlatouch_wshfl.so /tmp/build_wshfl/*.o
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "" /tmp/libvlatouch_wshfl.so; do
  echo "== lib=${lib:-product(dpp+permlane)}"
  VLATOUCH_LIB=$lib python bench.py --batch 1 --streams 1 --no-cpu-baseline --alt-compute-steps 0 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('full b1', d['value'], d['ms_per_step'])"
  VLATOUCH_LIB=$lib python bench.py --workload pi_refine --streams 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pi_refine s1', d['value'], d['ms_per_step'])"
  VLATOUCH_LIB=$lib python bench.py --workload dino_mlp --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('dino_mlp', d['value'], d['ms_per_step'])"
done; done
