cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1; tail -3 gpurun_out/r05_gpu_tests.log
bash tools/bench_all.sh r05 2>&1 | tail -20
bash tools/profile_round.sh r05 > gpurun_out/r05_profile.log 2>&1; tail -12 gpurun_out/r05_profile.log
bash tools/prof_b1.sh r05 > /dev/null 2>&1; head -12 gpurun_out/r05_per_step_b1.txt | cut -c1-110
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3
