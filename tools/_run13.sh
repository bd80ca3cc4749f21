cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/vla-touch_amd
O=gpurun_out; mkdir -p $O
prof() { local tag=$1 steps=$2; shift 2
  mkdir -p $O/prof_$tag
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python bench.py --steps $steps --warmup 2 --no-cpu-baseline "$@" > $O/r3l_$tag.json 2> $O/r3l_$tag.err
  cp $(find $O/prof_$tag -name "*.db" | head -1) $O/r3l_$tag.db; rm -rf $O/prof_$tag; }
prof siga 2 --workload siglip
prof sigb 6 --workload siglip
python tools/prof_per_step.py $O/r3l_siga.db 2 $O/r3l_sigb.db 6 > $O/r3l_per_step_siglip.txt
prof dina 5 --workload dino_mlp
prof dinb 25 --workload dino_mlp
python tools/prof_per_step.py $O/r3l_dina.db 5 $O/r3l_dinb.db 25 > $O/r3l_per_step_dino.txt
prof b1a 4 --streams 1 --batch 1
prof b1b 24 --streams 1 --batch 1
python tools/prof_per_step.py $O/r3l_b1a.db 4 $O/r3l_b1b.db 24 > $O/r3l_per_step_b1.txt
rm -f $O/r3l_*.db
head -14 $O/r3l_per_step_siglip.txt | cut -c1-150; head -16 $O/r3l_per_step_dino.txt | cut -c1-150; head -30 $O/r3l_per_step_b1.txt | cut -c1-150
