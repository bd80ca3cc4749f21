#!/bin/bash
# round-3 scratch: headline numbers after the fused U-Net path
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/r3s_full.json 2> $O/r3s_full.err
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams 1 > $O/r3s_full_s1.json 2>> $O/r3s_full.err
python bench.py --batch 1 --streams 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/r3s_b1.json 2>> $O/r3s_full.err
python bench.py --workload robot --batch 1 --streams 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/r3s_robot_b1.json 2>> $O/r3s_full.err
python bench.py --workload pi_refine --steps 20 --warmup 3 --no-cpu-baseline > $O/r3s_pi.json 2>> $O/r3s_full.err
python - <<'PY'
import json
for n in ("r3s_full","r3s_full_s1","r3s_b1","r3s_robot_b1","r3s_pi"):
    try:
        d=json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
    except Exception as e: print(n, "ERR", e)
PY
