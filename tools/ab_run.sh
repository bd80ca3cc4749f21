#!/bin/bash
# usage: tools/ab_run.sh TAG "ENV=VAL ..." [bench args]   -> gpurun_out/ab_TAG.{json,txt}: bench line + per-kernel rocprofv3 summary
TAG=$1; ENVS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$TAG
env $ENVS rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o ab -- python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_$TAG.json 2> gpurun_out/ab_$TAG.err
DB=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
python tools/prof_summary.py $DB > gpurun_out/ab_$TAG.txt 2>> gpurun_out/ab_$TAG.err
rm -rf gpurun_out/prof_$TAG
python - <<P
import json
try:
    d=json.loads(open("gpurun_out/ab_$TAG.json").read().strip().splitlines()[-1]); print("$TAG", d["value"], d["ms_per_step"])
except Exception as e: print("$TAG bench failed", e)
P
head -14 gpurun_out/ab_$TAG.txt
