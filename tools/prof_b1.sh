#!/bin/bash
# per-step kernel profile of the full pipeline at batch 1 (one chunk at a time: the latency case) -> gpurun_out/$1_per_step_b1.txt
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
A="--batch 1 --streams 1 --no-cpu-baseline --alt-compute-steps 0 --warmup 2"
rocprofv3 --kernel-trace --stats -d $O/prof_b1a -o a -- python bench.py $A --steps 10 > /dev/null 2> $O/${R}_prof_b1.err
rocprofv3 --kernel-trace --stats -d $O/prof_b1b -o b -- python bench.py $A --steps 50 > $O/${R}_prof_b1.json 2>> $O/${R}_prof_b1.err
python tools/prof_per_step.py $(find $O/prof_b1a -name "*.db" | head -1) 10 $(find $O/prof_b1b -name "*.db" | head -1) 50 > $O/${R}_per_step_b1.txt
rm -rf $O/prof_b1a $O/prof_b1b
head -45 $O/${R}_per_step_b1.txt
