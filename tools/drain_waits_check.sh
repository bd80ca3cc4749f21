#!/bin/bash
# a debug build of the library whose hand-counted `s_waitcnt vmcnt(N)` (gemm_pt / gemm_pp256d / gemm_pw / gemm_pws) all drain the queue, compared
# bit for bit with the product build on tools/drain_waits_check.py's cases (each run twice).   gpurun -- tools/drain_waits_check.sh
cd $GRAFT_REPO_ROOT/vla-touch_amd/csrc
rm -rf /tmp/build_drain; mkdir -p /tmp/build_drain
for f in *.hip; do e=""; [ $f = vt_uconv.hip ] && e="-Xclang -target-feature -Xclang -packed-fp32-ops"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVLATOUCH_DRAIN_WAITS $e -c $f -o /tmp/build_drain/${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvlatouch_drain.so /tmp/build_drain/*.o
cd $GRAFT_REPO_ROOT
python tools/drain_waits_check.py > gpurun_out/drain_product.txt 2> gpurun_out/drain.err
VLATOUCH_LIB=/tmp/libvlatouch_drain.so python tools/drain_waits_check.py > gpurun_out/drain_debug.txt 2>> gpurun_out/drain.err
wc -l gpurun_out/drain_product.txt gpurun_out/drain_debug.txt
if diff gpurun_out/drain_product.txt gpurun_out/drain_debug.txt > gpurun_out/drain_diff.txt; then echo "DRAIN CHECK: all $(wc -l < gpurun_out/drain_product.txt) digests identical (counted waits == drained waits)"; else echo "DRAIN CHECK: DIFFERENT"; head -20 gpurun_out/drain_diff.txt; fi
# and run-to-run: the two repetitions of every case inside one run
python - <<P
import re
for f in ("gpurun_out/drain_product.txt", "gpurun_out/drain_debug.txt"):
    d = {}
    for l in open(f):
        k, h = l.rsplit(" ", 1)
        d.setdefault(re.sub(r" #\d$", "", k), set()).add(h.strip())
    bad = [k for k, v in d.items() if len(v) != 1]
    print(f, "repeatable" if not bad else "NOT repeatable: %s" % bad)
P
