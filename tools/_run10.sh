cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
mkdir -p gpurun_out
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
bash tools/bench_all.sh r03 > gpurun_out/r03_bench_all.log 2>&1
cat gpurun_out/r03_bench_all.log; tail -30 gpurun_out/r03_profile_round.log
