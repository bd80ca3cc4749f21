#!/usr/bin/env python
"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):
    MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) x 1024 SIMDs)    (rocprofv3's own derived-metric formula)
GRBM_GUI_ACTIVE as reported here is summed over the 8 XCDs (a per-dispatch value ~8x the kernel's cycles), so it is divided by 8.
    python tools/pmc_mfma_summary.py gpurun_out/pmc/pmc_MFMA_counter_collection.csv"""
import csv, re, sys
from collections import defaultdict

agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        a = agg[name]
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a[0] += 1; a[1] += v
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a[2] += v
            a[3] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
SIMDS, XCDS = 1024, 8
rows = sorted(((a[3], k, a) for k, a in agg.items() if a[1] > 0), reverse=True)
print("# per kernel: dispatches, total ms, MfmaUtil % = MFMA_BUSY_CYCLES / (GUI_ACTIVE/8 x 1024 SIMDs), effective clock GHz = GUI_ACTIVE/8 / wall")
print("# NOTE on `clk`: GRBM_GUI_ACTIVE counts from before the first wave starts to after the last one retires, the kernel-trace wall time")
print("#      does not: for launches under ~100 us the quotient exceeds the chip's 2.4 GHz maximum and is NOT a clock (marked '~').  Only")
print("#      the long kernels' figure (the ~2 ms fused K|V projection) is an effective clock; MfmaUtil is a ratio of two counters and is")
print("#      unaffected.")
for wall, k, a in rows[:14]:
    gui = a[2] / XCDS
    short = wall / max(a[0], 1) < 100e3        # average launch under 100 us: the clock column is not meaningful
    print(f"{k[:78]:78s} n={a[0]:5d} {wall/1e6:9.2f} ms  MfmaUtil={100*a[1]/(gui*SIMDS):5.1f}%  clk={'~' if short else ' '}{gui/wall:4.2f} GHz")
