#!/bin/bash
# every workload of bench.py once -> gpurun_out/rNN_bench_*.json (copy into profiles/)
R=${1:-r02}; O=gpurun_out
run() { name=$1; shift; python bench.py "$@" > $O/${R}_bench_$name.json 2> $O/${R}_bench_$name.err; python - <<P
import json
try:
    d=json.loads(open("$O/${R}_bench_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["unit"], d["ms_per_step"], "p50", d["p50_step_latency_ms"])
except Exception as e: print("$name FAILED", e)
P
}
run full_n1
run full_streams2_n1 --streams 2 --no-cpu-baseline
run full_streams1_n1 --streams 1 --no-cpu-baseline
run full_streams4_n1 --streams 4 --no-cpu-baseline
run full_rmsvar_n1 --rms-mode var --no-cpu-baseline
run full_bf16act_n1 --rdt-compute bf16 --no-cpu-baseline
VLATOUCH_ATTN_FIXEDMAX=0 run full_online_softmax_n1 --no-cpu-baseline
run full_tactile64_n1 --force-dim 64 --no-cpu-baseline
run pi_refine_n1 --workload pi_refine --no-cpu-baseline
run pi_refine_streams1_n1 --workload pi_refine --no-cpu-baseline --streams 1
run dino_mlp_n1 --workload dino_mlp --no-cpu-baseline
run rdt_n1 --workload rdt --no-cpu-baseline
run rdt50_b16_n1 --workload rdt --rdt-steps 50 --batch 16 --steps 6 --warmup 1 --no-cpu-baseline
run siglip_n1 --workload siglip --no-cpu-baseline --steps 6 --warmup 1
run lstm_n1 --workload lstm --no-cpu-baseline
run marker_n1 --workload marker --no-cpu-baseline
run full_b1_n1 --batch 1 --streams 1 --no-cpu-baseline
run robot_n1 --workload robot --no-cpu-baseline --steps 6 --warmup 1
run robot_b1_n1 --workload robot --batch 1 --streams 1 --no-cpu-baseline --steps 10 --warmup 2
