cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe) > $O/r3f_tr_probe.txt 2>&1
timeout 600 python tools/gemm_bench_pw.py --abl > $O/r3f_abl.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_api.py -q --timeout=600 -m gpu > $O/r3f_tests.txt 2>&1
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3f_bench.json 2> $O/r3f_bench.err
head -40 $O/r3f_tr_probe.txt; grep ablation $O/r3f_abl.txt; tail -4 $O/r3f_tests.txt; cut -c1-200 $O/r3f_bench.json
