#!/bin/bash
# the headline command several times back to back on one box (run-to-run spread) + smoke() -> gpurun_out/$1_headline_repeats.txt
R=${1:-r06}; O=gpurun_out/${R}_headline_repeats.txt; : > $O
for i in 1 2 3; do
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('full x3 in flight', d['value'], 'chunks/s', d['ms_per_step'], 'ms/step | one at a time', d['latency_mode']['chunks_per_s'], '| bf16 activations', d['alt_rdt_compute']['chunks_per_s'], '| roofline', d['roofline']['frac'], [r['frac'] for r in d['roofline_other']], '| range flag', d['range_guard']['rdt_flag'], d['range_guard']['rdt_compute_used'])" >> $O
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 >> $O
cat $O
