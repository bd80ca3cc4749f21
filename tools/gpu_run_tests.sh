#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/gpu_run_tests.sh [pytest args]
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/gpu.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
