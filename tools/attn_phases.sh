#!/bin/bash
# bench build on the box, then tools/attn_phases.py; the product library is restored afterwards
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
cp ../vlatouch/libvlatouch_hip.so /tmp/libvlatouch_hip.product.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_BENCH_BUILD -c vt_attn.hip -o /tmp/vt_attn_bench.o 2>/dev/null
O=$(ls build/*.o | grep -v vt_attn.o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vlatouch/libvlatouch_hip.so $O /tmp/vt_attn_bench.o
cd $GRAFT_REPO_ROOT
python tools/attn_phases.py 2>&1 | grep -v amdgpu.ids
cp /tmp/libvlatouch_hip.product.so vla-touch_amd/vlatouch/libvlatouch_hip.so
