cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_rdt.py tests/test_gpu_fullsize.py tests/test_siglip.py tests/test_gpu_models.py -q --timeout=600 -m gpu > $O/r3i_tests.txt 2>&1
timeout 300 python tools/gemm_bench.py > $O/r3i_gemm.txt 2>&1
timeout 400 python bench.py --workload siglip --steps 6 --warmup 2 --no-cpu-baseline > $O/r3i_siglip.json 2> $O/r3i_siglip.err
timeout 400 python bench.py --workload dino_mlp --steps 20 --warmup 3 --no-cpu-baseline > $O/r3i_dino.json 2> $O/r3i_dino.err
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3i_bench.json 2> $O/r3i_bench.err
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/r3i_bench_s1.json 2>> $O/r3i_bench.err
tail -6 $O/r3i_tests.txt; cat $O/r3i_gemm.txt; cut -c1-200 $O/r3i_siglip.json $O/r3i_dino.json $O/r3i_bench.json $O/r3i_bench_s1.json
python - <<P
import json
for f in ("r3i_bench","r3i_bench_s1"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["roofline"]["achieved"], d["roofline"]["avg_launch_us"], [ (r["kernel"][:20], r["achieved"], r["avg_launch_us"]) for r in d.get("roofline_other",[])])
    except Exception as e: print(f, "failed", e)
P
