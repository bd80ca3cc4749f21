#!/bin/bash
# the round-end GPU call: (pytest -m gpu unless SKIP_TESTS=1), every bench workload, the rocprofv3 passes, the batch-1 per-step table, smoke()
#   tools/final_run.sh r06
R=${1:-r06}
cd $GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then python -m pytest tests -x -q -m gpu > gpurun_out/${R}_gpu_tests.log 2>&1; tail -3 gpurun_out/${R}_gpu_tests.log; fi
bash tools/bench_all.sh $R 2>&1 | tail -20
bash tools/profile_round.sh $R > gpurun_out/${R}_profile.log 2>&1; tail -12 gpurun_out/${R}_profile.log
bash tools/prof_b1.sh $R > /dev/null 2>&1; head -12 gpurun_out/${R}_per_step_b1.txt | cut -c1-110
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3
