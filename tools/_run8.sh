cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -q --timeout=900 -m gpu > $O/r3h_tests.txt 2>&1
timeout 600 python bench.py --workload robot --steps 6 --warmup 2 --no-cpu-baseline > $O/r3h_robot.json 2> $O/r3h_robot.err
tail -12 $O/r3h_tests.txt; cut -c1-300 $O/r3h_robot.json; tail -3 $O/r3h_robot.err
