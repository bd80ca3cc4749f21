#!/bin/bash
# round-3 scratch: small per-step Linears on the packed small-M tile
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_rdt.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3n_tests.txt 2>&1
tail -3 gpurun_out/r3n_tests.txt
python bench.py --batch 1 --streams 1 --steps 10 --warmup 2 > gpurun_out/r3n_b1.json 2> gpurun_out/r3n_b1.err
python bench.py --steps 10 --warmup 2 > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err
python bench.py --workload robot --batch 1 --streams 1 --steps 10 --warmup 2 > gpurun_out/r3n_robot_b1.json 2> gpurun_out/r3n_robot_b1.err
python - <<'PY'
import json
for n in ("r3n_b1","r3n_bench","r3n_robot_b1"):
    try:
        d=json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
    except Exception as e: print(n, "ERR", e)
PY
