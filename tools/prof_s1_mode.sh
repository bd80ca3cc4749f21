#!/bin/bash
# per-step kernel table of the full pipeline, one batch at a time, for one set of extra bench flags:  tools/prof_s1_mode.sh TAG [bench flags...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o s1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --latency-steps 0 "$@" > /dev/null 2> $O/${TAG}_prof.err
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}b -o s1b -- python bench.py --steps 13 --warmup 2 --no-cpu-baseline --streams 1 --latency-steps 0 "$@" > /dev/null 2>> $O/${TAG}_prof.err
python tools/prof_per_step.py $(find $O/prof_$TAG -name "*.db" | head -1) 5 $(find $O/prof_${TAG}b -name "*.db" | head -1) 13 > $O/${TAG}_per_step_streams1.txt
rm -rf $O/prof_$TAG $O/prof_${TAG}b
head -14 $O/${TAG}_per_step_streams1.txt
