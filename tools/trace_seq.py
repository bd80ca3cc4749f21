#!/usr/bin/env python
"""Per-launch sequence of the LAST bench step from a rocprofv3 --kernel-trace csv:
    python tools/trace_seq.py <kernel_trace.csv> [first_kernel_substring] [max_rows]
Prints start offset, duration, gap to the previous launch's end, grid / workgroup size and the kernel name, so a dependent
launch chain (the pi_I U-Nets, the per-denoise-step RDT launches) can be read launch by launch."""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    anchor = sys.argv[2] if len(sys.argv) > 2 else "sinusoid_kernel<float>"
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if not idx:
        raise SystemExit(f"no launch matches {anchor!r}")
    # the last-but-one anchor .. the last anchor = one period of the chain
    a, b = (idx[-2], idx[-1]) if len(idx) > 1 else (idx[-1], len(rows))
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    tot = 0
    print(f"# one period between the last two launches of {anchor!r}: {b - a} launches, {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
    print(f"{'t_us':>9} {'dur_us':>8} {'gap_us':>7} {'grid':>7} {'wg':>5}  kernel")
    for r in rows[a:b][:nmax]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"unsigned short", "bf16", name)
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {(s - prev_end) / 1e3:7.2f} {grid // max(wg, 1):7d} {wg:5d}  {name[:110]}")
        prev_end = e
        tot += e - s
    print(f"# busy {tot / 1e3:.1f} us")


if __name__ == "__main__":
    main()
