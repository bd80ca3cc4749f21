#!/bin/bash
# Fabric traffic (PMC FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and time of the fused K|V projection (139 968 x 4096 x 2048) under the tile-walk / store-policy
# variants of csrc/vt_gemm_pt.hip.  Bench build (the store-policy bits live there); the product library is restored afterwards.
#   gpurun -- tools/kv_traffic.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
rm -rf build_bench; mkdir -p build_bench
for f in *.hip; do o=build_bench/${f%.hip}.o; e=""; [ $f = vt_uconv.hip ] && e="-Xclang -target-feature -Xclang -packed-fp32-ops"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_BENCH_BUILD $e -c $f -o $o 2>/dev/null & done; wait
cp ../vlatouch/libvlatouch_hip.so /tmp/libvlatouch_hip.product.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vlatouch/libvlatouch_hip.so build_bench/*.o
cd $GRAFT_REPO_ROOT
O=gpurun_out/kvt; mkdir -p $O
one() {   # one TAG "ENV=VAL ..."
  TAG=$1; ENVS=$2
  T=$(env $ENVS python tools/gemm_bench_pt.py kv 2>&1 | grep "K|V" | sed 's/.*pt *\([0-9.]*\) us.*/\1/')
  for C in FETCH_SIZE WRITE_SIZE; do
    env $ENVS rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$TAG -o $C -- python tools/gemm_bench_pt.py kv > /dev/null 2>&1
  done
  python - <<P
import csv, glob
def avg(c):
    f = glob.glob("$O/$TAG/**/%s_counter_collection.csv" % c, recursive=True)
    if not f: return float("nan")
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c and "gemm_pt_kernel" in r["Kernel_Name"]]
    return sum(v) / max(len(v), 1) * 1024
rd, wr = 2 * avg("FETCH_SIZE"), avg("WRITE_SIZE")
print("%-34s %8s us   read %.2f GB  write %.2f GB  total %.2f GB  (algorithmic 0.59 + 1.15)" % ("$TAG", "$T", rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
P
  rm -rf $O/$TAG
}
if [ "$1" = "ablate" ]; then      # timing-only: does the time follow the traffic?  A panel L2-resident (reads drop to the W stream), then also without stores
  one default ""
  one A_panel_resident "VLATOUCH_PT_ABL=128"
  one A_resident_no_stores "VLATOUCH_PT_ABL=136"
  one no_stores "VLATOUCH_PT_ABL=8"
  one default_again ""
else
one default ""
one store_nt "VLATOUCH_PT_ABL=16"
one store_sc0nt "VLATOUCH_PT_ABL=32"
one store_sc0 "VLATOUCH_PT_ABL=64"
one both_halves_per_block "VLATOUCH_PT_KV_SPLIT=0"
one both_halves_nt "VLATOUCH_PT_KV_SPLIT=0 VLATOUCH_PT_ABL=16"
one gm2 "VLATOUCH_GEMM_GM=2"
one gm8 "VLATOUCH_GEMM_GM=8"
one gm8_both "VLATOUCH_GEMM_GM=8 VLATOUCH_PT_KV_SPLIT=0"
one default_again ""
fi
cp /tmp/libvlatouch_hip.product.so vla-touch_amd/vlatouch/libvlatouch_hip.so
