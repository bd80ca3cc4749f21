#!/bin/bash
# A/B of the nt cache policy on the small-M weight stream (vt_gemm_pws.hip) and on the U-Net weight fragments (vt_uconv.hip): variants built on the box
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc"
cp ../vlatouch/libvlatouch_hip.so /tmp/lib_base.so
/opt/rocm/bin/hipcc $F -DVLATOUCH_PWS_NT -c vt_gemm_pws.hip -o /tmp/pws_nt.o 2>/dev/null &
/opt/rocm/bin/hipcc $F -DVLATOUCH_UCONV_NT -Xclang -target-feature -Xclang -packed-fp32-ops -c vt_uconv.hip -o /tmp/uconv_nt.o 2>/dev/null &
wait
O=$(ls build/*.o | grep -v vt_gemm_pws.o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_P.so $O /tmp/pws_nt.o
O=$(ls build/*.o | grep -v vt_uconv.o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_U.so $O /tmp/uconv_nt.o
cd $GRAFT_REPO_ROOT
one() { python bench.py $* --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for v in base P U; do
  cp /tmp/lib_$v.so vla-touch_amd/vlatouch/libvlatouch_hip.so
  echo "== $v full b1:        $(one --batch 1 --streams 1)"
  echo "== $v pi_refine b1:   $(one --workload pi_refine --batch 1 --streams 1)"
  echo "== $v pi_refine s1:   $(one --workload pi_refine --streams 1)"
  echo "== $v full b4:        $(one --batch 4 --streams 1)"
done; done
cp /tmp/lib_base.so vla-touch_amd/vlatouch/libvlatouch_hip.so
