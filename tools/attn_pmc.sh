#!/bin/bash
# SQ counters of the ViT self-attention kernel on tools/attn_bench.py (two --pmc passes, kernel trace only): wave-cycle shares, LDS conflicts, instruction mix
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O/pmc_attn
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_attn -o a -- python tools/attn_bench.py > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find $O/pmc_attn -name "a_counter_collection.csv" | head -1) | grep -i "attn\|^#"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_attn -o b -- python tools/attn_bench.py > /dev/null 2>&1
python - <<P
import csv, re, glob
from collections import defaultdict
f = glob.glob("$O/pmc_attn/**/b_counter_collection.csv", recursive=True)[0]
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, a in agg.items():
    if "attn" not in k: continue
    w = max(a["SQ_WAVES"], 1)
    print(f"{k[:60]:60s} n={n[k]} per wave: VALU {a['SQ_INSTS_VALU']/w:7.0f} MFMA {a['SQ_INSTS_MFMA']/w:6.0f} LDS {a['SQ_INSTS_LDS']/w:6.0f} SALU {a['SQ_INSTS_SALU']/w:6.0f} VMEM_RD {a['SQ_INSTS_VMEM_RD']/w:5.0f} | waves {w/n[k]:.0f} busy_cycles/launch {a['SQ_BUSY_CYCLES']/n[k]:.0f} active_valu/launch {a['SQ_ACTIVE_INST_VALU']/n[k]:.0f}")
P
rm -rf $O/pmc_attn
