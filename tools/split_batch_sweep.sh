#!/bin/bash
# Round 6: does a 32-episode batch run faster as two 16-episode halves on two streams (pipelining inside one batch)?  full pipeline, same box
set -u
out=gpurun_out/split; mkdir -p $out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --alt-compute-steps 0 --latency-steps 0 "$@" > $out/$name.json 2> $out/$name.err; python - <<P
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], "chunks/s", d["ms_per_step"], "ms/step p50", d["p50_step_latency_ms"])
except Exception as e: print("$name FAILED", e)
P
}
run b32_s1 --batch 32 --streams 1
run b16_s2 --batch 16 --streams 2 --steps 40
run b8_s4 --batch 8 --streams 4 --steps 80
run b32_s3 --batch 32 --streams 3
run b16_s6 --batch 16 --streams 6 --steps 40
run b16_s4 --batch 16 --streams 4 --steps 40
