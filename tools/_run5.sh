cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
timeout 600 python tools/gemm_bench_pw.py > $O/r3e_pws.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_rdt.py tests/test_gpu_fullsize.py tests/test_gpu_api.py tests/test_marker.py tests/test_gpu_primitives.py -q --timeout=600 -m gpu > $O/r3e_tests.txt 2>&1
prof() { # tag steps args...
  local tag=$1 steps=$2; shift 2
  mkdir -p $O/prof_$tag
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python bench.py --steps $steps --warmup 2 --no-cpu-baseline "$@" > $O/r3e_$tag.json 2> $O/r3e_$tag.err
  cp $(find $O/prof_$tag -name "*.db" | head -1) $O/r3e_$tag.db
  rm -rf $O/prof_$tag
}
prof s1a 4 --streams 1
prof s1b 12 --streams 1
python tools/prof_per_step.py $O/r3e_s1a.db 4 $O/r3e_s1b.db 12 > $O/r3e_per_step_s1.txt
prof b1a 4 --streams 1 --batch 1
prof b1b 24 --streams 1 --batch 1
python tools/prof_per_step.py $O/r3e_b1a.db 4 $O/r3e_b1b.db 24 > $O/r3e_per_step_b1.txt
prof siga 2 --workload siglip
prof sigb 6 --workload siglip
python tools/prof_per_step.py $O/r3e_siga.db 2 $O/r3e_sigb.db 6 > $O/r3e_per_step_siglip.txt
rm -f $O/r3e_*.db
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r3e_bench.json 2> $O/r3e_bench.err
head -20 $O/r3e_pws.txt; tail -8 $O/r3e_tests.txt; cut -c1-220 $O/r3e_bench.json $O/r3e_b1b.json $O/r3e_sigb.json; head -50 $O/r3e_per_step_s1.txt; head -40 $O/r3e_per_step_b1.txt; head -30 $O/r3e_per_step_siglip.txt
