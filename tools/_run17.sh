#!/bin/bash
# round-3 scratch: fused U-Net — parity tests, launch-by-launch trace at B=32 and B=1, pi_refine timings
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "unet or si_sample" 2>&1 | tail -5
for B in 32 1; do
  rm -rf $O/tr
  rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python bench.py --workload pi_refine --batch $B --streams 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/r3r_pi_b$B.json 2> $O/r3r_pi_b$B.err
  F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
  python tools/trace_seq.py $F "ufinal_kernel" 200 > $O/r3r_seq_pi_b$B.txt
  timeout 300 python bench.py --workload pi_refine --batch $B --streams 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/r3q_pi_b$B.json 2> $O/r3q_pi_b$B.err
done
rm -rf $O/tr
python - <<'PY'
import json
for n in ("r3q_pi_b32","r3q_pi_b1"):
    try:
        d=json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
head -40 $O/r3r_seq_pi_b32.txt
