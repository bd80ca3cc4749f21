cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_siglip.py tests/test_gpu_models.py tests/test_gpu_rdt.py -q --timeout=600 -m gpu > $O/r3g_tests.txt 2>&1
timeout 600 python tools/gemm_bench_pw.py --big > $O/r3g_pw.txt 2>&1
timeout 400 python bench.py --workload siglip --steps 6 --warmup 2 --no-cpu-baseline > $O/r3g_siglip.json 2> $O/r3g_siglip.err
VLATOUCH_ATTN16=0 timeout 400 python bench.py --workload siglip --steps 6 --warmup 2 --no-cpu-baseline > $O/r3g_siglip_old.json 2>> $O/r3g_siglip.err
timeout 400 python bench.py --workload dino_mlp --steps 20 --warmup 3 --no-cpu-baseline > $O/r3g_dino.json 2> $O/r3g_dino.err
VLATOUCH_ATTN16=0 timeout 400 python bench.py --workload dino_mlp --steps 20 --warmup 3 --no-cpu-baseline > $O/r3g_dino_old.json 2>> $O/r3g_dino.err
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3g_bench.json 2> $O/r3g_bench.err
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/r3g_bench_s1.json 2>> $O/r3g_bench.err
tail -6 $O/r3g_tests.txt; grep "timing" -A12 $O/r3g_pw.txt; cut -c1-200 $O/r3g_siglip.json $O/r3g_siglip_old.json $O/r3g_dino.json $O/r3g_dino_old.json $O/r3g_bench.json $O/r3g_bench_s1.json
