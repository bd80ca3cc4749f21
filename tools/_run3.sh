cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 600 python tools/gemm_bench_pw.py > $O/r3c_pws.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_rdt.py tests/test_gpu_fullsize.py -q -x --timeout=600 -m gpu > $O/r3c_tests.txt 2>&1
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3c_bench.json 2> $O/r3c_bench.err
VLATOUCH_ATTN_FIXEDMAX=0 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3c_bench_nofix.json 2>> $O/r3c_bench.err
mkdir -p $O/prof_b1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_b1 -o b1 -- python bench.py --batch 1 --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/r3c_b1.json 2> $O/r3c_b1.err
python tools/prof_summary.py $(find $O/prof_b1 -name "*.db" | head -1) 10 > $O/r3c_b1_kernel_stats.txt
rm -rf $O/prof_b1
VLATOUCH_PWS=0 timeout 400 python bench.py --batch 1 --streams 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/r3c_b1_old.json 2>> $O/r3c_b1.err
grep -v "^  check\|^  M=" $O/r3c_pws.txt | tail -40; tail -5 $O/r3c_tests.txt; cut -c1-200 $O/r3c_bench.json $O/r3c_bench_nofix.json $O/r3c_b1.json $O/r3c_b1_old.json; python - <<P
import json
for f in ("r3c_bench","r3c_bench_nofix"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], [ (r["kernel"][:20], r["achieved"], r["avg_launch_us"]) for r in d.get("roofline_other",[])])
    except Exception as e: print(f, "failed", e)
P
head -16 $O/r3c_b1_kernel_stats.txt
