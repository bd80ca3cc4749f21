#!/usr/bin/env python
"""HBM traffic of the dominant kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-B requests at 64 B -> doubled for wide coalesced reads.
    python tools/pmc_summary.py gpurun_out/pmc/pmc_FETCH_SIZE_counter_collection.csv gpurun_out/pmc/pmc_WRITE_SIZE_counter_collection.csv [profiles/pmc_traffic.json]"""
import csv, sys, re
from collections import defaultdict

def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            a = agg[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# per kernel: dispatches, HBM read GB (FETCH_SIZE KB x 2 gfx950 correction), HBM write GB (WRITE_SIZE KB), per-dispatch MB")
rows = []
for k in fetch:
    n, f = fetch[k]
    w = write.get(k, [0, 0.0])[1]
    rows.append((2 * f * 1024 + w * 1024, k, n, 2 * f * 1024, w * 1024))
for tot, k, n, rd, wr in sorted(rows, reverse=True)[:12]:
    print(f"{k[:70]:70s} n={n:5d} read={rd/1e9:8.2f} GB write={wr/1e9:8.2f} GB  per-dispatch={(rd+wr)/n/1e6:9.2f} MB")
import json
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes), bench.py --steps 1 --warmup 1 --no-graph; bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction)"}
# class "gemm_pp256" = the 256-square ping-pong tile in both forms (vt_prof class 2): gemm_pp256d_kernel and the persistent gemm_pt_kernel
for cls in ("gemm_pp256", "gemm_ppk", "gemm_pw_", "gemm_pws", "gemm_glds", "attn_kvt", "gemm_pt_kernel<unsigned short, 0"):
    if cls.startswith("gemm_pt_kernel"):      # the fused K|V projection in either 16-bit type: gemm_pt_kernel<unsigned short | half_t, 0, 0>
        g = [r for r in rows if r[1].startswith("gemm_pt_kernel<") and r[1].rstrip().endswith(", 0, 0>")]
    else:
        g = [r for r in rows if cls in r[1] or (cls == "gemm_pp256" and "gemm_pt_kernel" in r[1])]
    tot = sum(r[0] for r in g); n = sum(r[2] for r in g)
    print(f"CLASS {cls}: dispatches={n} bytes={tot:.0f} per_launch_bytes={tot/max(n,1):.0f}")
    out["gemm_pt_kv" if cls.startswith("gemm_pt_kernel") else cls.rstrip("_")] = {"dispatches": n, "per_launch_bytes": tot / max(n, 1)}
if len(sys.argv) > 3:
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vla-touch_amd"))
    from vlatouch import _lib
    out["csrc_sha16"] = _lib.csrc_sha16()          # which kernel sources the counters were taken on (bench.py compares it with the tree it runs from)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
