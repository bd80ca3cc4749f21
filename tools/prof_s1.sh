#!/bin/bash
# per-step kernel table of the full pipeline, one batch at a time (difference of a 13-step and a 5-step trace) -> gpurun_out/$1_per_step_streams1.txt
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o s1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --alt-compute-steps 0 --streams 1 > /dev/null 2> $O/${R}_prof_s1.err
rocprofv3 --kernel-trace --stats -d $O/prof_s1b -o s1b -- python bench.py --steps 13 --warmup 2 --no-cpu-baseline --alt-compute-steps 0 --streams 1 > /dev/null 2>> $O/${R}_prof_s1.err
python tools/prof_per_step.py $(find $O/prof_s1 -name "*.db" | head -1) 5 $(find $O/prof_s1b -name "*.db" | head -1) 13 > $O/${R}_per_step_streams1.txt
rm -rf $O/prof_s1 $O/prof_s1b
head -16 $O/${R}_per_step_streams1.txt
