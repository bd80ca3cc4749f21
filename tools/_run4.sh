cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 600 python tools/gemm_bench_pw.py > $O/r3d_pws.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_rdt.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -q -x --timeout=600 -m gpu > $O/r3d_tests.txt 2>&1
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3d_bench.json 2> $O/r3d_bench.err
mkdir -p $O/prof_b1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_b1 -o b1 -- python bench.py --batch 1 --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/r3d_b1.json 2> $O/r3d_b1.err
python tools/prof_summary.py $(find $O/prof_b1 -name "*.db" | head -1) 10 > $O/r3d_b1_kernel_stats.txt
rm -rf $O/prof_b1
timeout 400 python bench.py --batch 1 --streams 1 --steps 20 --warmup 2 --no-cpu-baseline > $O/r3d_b1_20.json 2>> $O/r3d_b1.err
cat $O/r3d_pws.txt | tail -50; tail -5 $O/r3d_tests.txt; cut -c1-200 $O/r3d_bench.json $O/r3d_b1.json $O/r3d_b1_20.json; head -16 $O/r3d_b1_kernel_stats.txt
