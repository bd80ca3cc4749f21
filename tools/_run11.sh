cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_siglip.py tests/test_gpu_models.py -q --timeout=600 -m gpu > $O/r3j_tests.txt 2>&1
timeout 400 python bench.py --workload siglip --steps 6 --warmup 2 --no-cpu-baseline > $O/r3j_siglip.json 2> $O/r3j_siglip.err
VLATOUCH_ROWNORM_WAVE=0 timeout 400 python bench.py --workload siglip --steps 6 --warmup 2 --no-cpu-baseline > $O/r3j_siglip_blocknorm.json 2>> $O/r3j_siglip.err
timeout 400 python bench.py --workload dino_mlp --steps 20 --warmup 3 --no-cpu-baseline > $O/r3j_dino.json 2> $O/r3j_dino.err
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s --timeout=900 -m gpu 2>&1 | grep "RDT-1B\|passed\|failed\|chained" > $O/r3j_fullsize.txt
tail -6 $O/r3j_tests.txt; cut -c1-200 $O/r3j_siglip.json $O/r3j_siglip_blocknorm.json $O/r3j_dino.json; cat $O/r3j_fullsize.txt
