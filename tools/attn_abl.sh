#!/bin/bash
# Timing-only ablations of the ViT self-attention kernel (bench build; garbage results): what bounds attn16_kernel?
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
rm -rf build_bench; mkdir -p build_bench
for f in *.hip; do o=build_bench/${f%.hip}.o; e=""; [ $f = vt_uconv.hip ] && e="-Xclang -target-feature -Xclang -packed-fp32-ops"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_BENCH_BUILD $e -c $f -o $o 2>/dev/null & done; wait
cp ../vlatouch/libvlatouch_hip.so /tmp/libvlatouch_hip.product.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vlatouch/libvlatouch_hip.so build_bench/*.o
cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 8 3 7 15; do echo "== VLATOUCH_ATTN_ABL=$a (1 no softmax, 2 no PV, 4 no QK^T, 8 no staging)"; VLATOUCH_ATTN_ABL=$a python tools/attn_bench.py 2>&1 | grep "TF/s"; done
cp /tmp/libvlatouch_hip.product.so vla-touch_amd/vlatouch/libvlatouch_hip.so
