#!/bin/bash
# round-3 scratch: launch-by-launch trace of the full step at B=1 (one RDT block + the tail)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
rm -rf $O/tr
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python bench.py --batch 1 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/r3t_b1.json 2> $O/r3t_b1.err
F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python tools/trace_seq.py $F "dpm_update_kernel" 400 > $O/r3t_seq_b1.txt
rm -rf $O/tr
head -120 $O/r3t_seq_b1.txt
