#!/bin/bash
# A/B of the store cache policy of the per-denoise-step Linears' epilogue (vt_gemm_pw.hip, -DVLATOUCH_PW_ST=1 nt | 2 sc0 sc1 write-through): variants built on the box
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc"
cp ../vlatouch/libvlatouch_hip.so /tmp/lib_0.so
for v in 1 2; do /opt/rocm/bin/hipcc $F -DVLATOUCH_PW_ST=$v -c vt_gemm_pw.hip -o /tmp/pw_st$v.o 2>/dev/null & done; wait
O=$(ls build/*.o | grep -v vt_gemm_pw.o)
for v in 1 2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$v.so $O /tmp/pw_st$v.o; done
cd $GRAFT_REPO_ROOT
one() { python bench.py $* --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for v in 0 1 2; do
  cp /tmp/lib_$v.so vla-touch_amd/vlatouch/libvlatouch_hip.so
  echo "== policy $v streams1: $(one --streams 1)   full: $(one)"
done; done
cp /tmp/lib_0.so vla-touch_amd/vlatouch/libvlatouch_hip.so
