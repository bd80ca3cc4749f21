#!/bin/bash
# Round 6: rows per launch for the per-denoise-step Linears — the full pipeline at 32 / 64 / 96 episodes per batch on ONE box.
# Each line carries roofline_other[gemm_pw].frac and latency_mode.  Output: gpurun_out/rows/*.json
set -u
out=gpurun_out/rows; mkdir -p $out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --alt-compute-steps 0 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(head -c 300 $out/$name.json)"; }
run b32_s1 --batch 32 --streams 1 --latency-steps 0
run b32_s3 --batch 32 --streams 3
run b64_s1 --batch 64 --streams 1 --latency-steps 0
run b64_s2 --batch 64 --streams 2
run b96_s1 --batch 96 --streams 1 --latency-steps 0
run b128_s1 --batch 128 --streams 1 --latency-steps 0
