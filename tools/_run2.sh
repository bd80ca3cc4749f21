cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 600 python tools/gemm_bench_pw.py > $O/r3b_pw.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_rdt.py tests/test_gpu_fullsize.py tests/test_marker.py tests/test_gpu_api.py -q -x --timeout=600 -m gpu > $O/r3b_tests.txt 2>&1
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3b_bench.json 2> $O/r3b_bench.err
mkdir -p $O/prof_s1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o s1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > $O/r3b_bench_s1.json 2>> $O/r3b_bench.err
python tools/prof_summary.py $(find $O/prof_s1 -name "*.db" | head -1) 10 > $O/r3b_kernel_stats_s1.txt
rm -rf $O/prof_s1
tail -12 $O/r3b_pw.txt; tail -5 $O/r3b_tests.txt; cut -c1-200 $O/r3b_bench.json $O/r3b_bench_s1.json; head -24 $O/r3b_kernel_stats_s1.txt
