cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 mfma16_probe.hip -o /tmp/mfma16_probe 2>/dev/null && /tmp/mfma16_probe) > $O/r3k_mfma16.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_siglip.py tests/test_gpu_models.py tests/test_gpu_rdt.py -q --timeout=600 -m gpu > $O/r3k_tests.txt 2>&1
timeout 400 python bench.py --workload siglip --steps 6 --warmup 2 --no-cpu-baseline > $O/r3k_siglip.json 2> $O/r3k_siglip.err
timeout 400 python bench.py --workload dino_mlp --steps 20 --warmup 3 --no-cpu-baseline > $O/r3k_dino.json 2> $O/r3k_dino.err
cat $O/r3k_mfma16.txt; grep -c "FAILED" $O/r3k_tests.txt; grep "FAILED" $O/r3k_tests.txt | head -12; tail -2 $O/r3k_tests.txt; cut -c1-200 $O/r3k_siglip.json $O/r3k_dino.json
