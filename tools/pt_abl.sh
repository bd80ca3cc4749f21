#!/bin/bash
# Timing-only ablations of the persistent K|V projection tile (bench build; results are garbage): what do the epilogue slots and each role cost?
#   gpurun -- tools/pt_abl.sh
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
rm -rf build_bench; mkdir -p build_bench
for f in *.hip; do o=build_bench/${f%.hip}.o; e=""; [ $f = vt_uconv.hip ] && e="-Xclang -target-feature -Xclang -packed-fp32-ops"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_BENCH_BUILD $e -c $f -o $o & done; wait
cp ../vlatouch/libvlatouch_hip.so /tmp/libvlatouch_hip.product.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vlatouch/libvlatouch_hip.so build_bench/*.o
cd $GRAFT_REPO_ROOT
for a in 0 16 32 64 0; do echo "== VLATOUCH_PT_ABL=$a"; VLATOUCH_PT_ABL=$a python tools/gemm_bench_pt.py kv 2>&1 | grep "K|V"; done
cp /tmp/libvlatouch_hip.product.so vla-touch_amd/vlatouch/libvlatouch_hip.so
