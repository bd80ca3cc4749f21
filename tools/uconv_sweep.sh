#!/bin/bash
# Sweep of the fused U-Net tile cost model (csrc/vt_unet_fused.hip pick_tile): us per k-step, us per consumer gather round, block cap.
#   gpurun -- 'bash tools/uconv_sweep.sh'      -> gpurun_out/uconv_sweep.txt
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; : > $O/uconv_sweep.txt
for B in 32 1; do
for TS in 0.05 0.1 0.2; do for TR in 0.5 1.5 3.0; do for BC in 256 512; do
  v=$(VLATOUCH_UC_TSTEP=$TS VLATOUCH_UC_TROUND=$TR VLATOUCH_UC_BLOCKS=$BC python bench.py --workload pi_refine --batch $B --streams 1 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "B=$B tstep=$TS tround=$TR blocks=$BC ms=$v" | tee -a $O/uconv_sweep.txt
done; done; done; done
