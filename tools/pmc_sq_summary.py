#!/usr/bin/env python
"""LDS bank-conflict ratio and wave-cycle breakdown per kernel from one rocprofv3 --pmc pass:
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv ...
    python tools/pmc_sq_summary.py <counter_collection.csv>
lds_conflict = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE; wait_any / wait_inst / active = share of SQ_WAVE_CYCLES parked on s_waitcnt or a barrier,
stalled at issue (MFMA pipe, dependencies), issuing."""
import csv, re, sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[name] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))
print("# " + __doc__.strip().splitlines()[1].strip())
print("# per kernel (sorted by wave cycles): dispatches, LDS bank-conflict cycles / LDS active cycles; share of wave cycles waiting / stalled at issue / issuing")
for k, a in rows[:16]:
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    lds = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(a.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    print(f"{k[:72]:72s} n={n[k]:5d} lds_conflict={lds:6.3f}  wait_any={a.get('SQ_WAIT_ANY', 0) / wc:5.2f}  wait_inst={a.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f}  active={a.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.2f}")
