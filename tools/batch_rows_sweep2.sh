#!/bin/bash
# Round 6: second part of the rows-per-launch sweep — 96-episode batches with 2 / 3 in flight, and the batch that fills the 160-row tile grid exactly (38 x 67 = 2546 rows = 16 m-tiles)
set -u
out=gpurun_out/rows; mkdir -p $out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --alt-compute-steps 0 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(head -c 200 $out/$name.json)"; }
run b96_s2 --batch 96 --streams 2
run b96_s3 --batch 96 --streams 3 --latency-steps 0
run b38_s1 --batch 38 --streams 1 --latency-steps 0
run b38_s3 --batch 38 --streams 3 --latency-steps 0
