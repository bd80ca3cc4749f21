#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench + PMC passes (each counter set in its own run, as
# MI355X_MICROARCH.md prescribes) -> gpurun_out/rNN_*.txt ; copy the ones to be judged into profiles/.
#   tools/profile_round.sh r02
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O/pmc
# 1. per-kernel time of the headline command line (2 batches in flight, hipGraph)
rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --alt-compute-steps 0 > $O/${R}_prof_bench.json 2> $O/${R}_prof_bench.err
python tools/prof_summary.py $(find $O/prof_full -name "*.db" | head -1) 10 > $O/${R}_full_kernel_stats.txt
rm -rf $O/prof_full
# 2. the same with one batch in flight (kernel durations without co-running kernels)
rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o s1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --alt-compute-steps 0 --streams 1 > $O/${R}_prof_bench_s1.json 2>> $O/${R}_prof_bench.err
python tools/prof_summary.py $(find $O/prof_s1 -name "*.db" | head -1) 10 > $O/${R}_full_kernel_stats_streams1.txt
cp $(find $O/prof_s1 -name "*.db" | head -1) $O/_s1_5.db
rm -rf $O/prof_s1
# 2b. the same with 13 steps: the difference of the two runs per step is free of set-up work (weight generation, packing, one-off copies)
rocprofv3 --kernel-trace --stats -d $O/prof_s1b -o s1b -- python bench.py --steps 13 --warmup 2 --no-cpu-baseline --alt-compute-steps 0 --streams 1 > /dev/null 2>> $O/${R}_prof_bench.err
python tools/prof_per_step.py $O/_s1_5.db 5 $(find $O/prof_s1b -name "*.db" | head -1) 13 > $O/${R}_per_step_streams1.txt
rm -rf $O/prof_s1b $O/_s1_5.db
# 3. PMC: HBM-side bytes (two passes), MFMA utilisation, SQ stalls — eager, one step
PB="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --alt-compute-steps 0 --no-graph --streams 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc -o pmc_$C -- $PB > /dev/null 2>> $O/${R}_prof_bench.err
done
python tools/pmc_summary.py $(find $O/pmc -name "pmc_FETCH_SIZE_counter_collection.csv" | head -1) $(find $O/pmc -name "pmc_WRITE_SIZE_counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/${R}_pmc_hbm_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -o pmc_MFMA -- $PB > /dev/null 2>> $O/${R}_prof_bench.err
python tools/pmc_mfma_summary.py $(find $O/pmc -name "pmc_MFMA_counter_collection.csv" | head -1) > $O/${R}_pmc_mfma_util.txt
rm -rf $O/pmc
head -20 $O/${R}_full_kernel_stats.txt; cat $O/${R}_pmc_hbm_traffic.txt | tail -8; head -12 $O/${R}_pmc_mfma_util.txt
