cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 600 python tools/gemm_bench_pw.py > $O/r3a_pw.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_rdt.py tests/test_gpu_fullsize.py -q -x --timeout=600 > $O/r3a_tests.txt 2>&1
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3a_bench.json 2> $O/r3a_bench.err
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/r3a_bench_s1.json 2>> $O/r3a_bench.err
VLATOUCH_PW=0 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/r3a_bench_s1_old.json 2>> $O/r3a_bench.err
mkdir -p $O/prof_b1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_b1 -o b1 -- python bench.py --batch 1 --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/r3a_b1.json 2> $O/r3a_b1.err
python tools/prof_summary.py $(find $O/prof_b1 -name "*.db" | head -1) 10 > $O/r3a_b1_kernel_stats.txt
rm -rf $O/prof_b1
tail -30 $O/r3a_pw.txt; tail -5 $O/r3a_tests.txt; cut -c1-200 $O/r3a_bench.json $O/r3a_bench_s1.json $O/r3a_bench_s1_old.json
