#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as text:
    python tools/prof_summary.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_name.txt
Per kernel: calls, total / average duration, share of GPU busy time (the `top_kernels` view of rocprofv3)."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
    print(f"# GPU busy time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches" + (f" ({steps} bench steps incl. warm-up/profiling legs)" if steps else ""))
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows[:40]:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"unsigned short", "bf16", name)
        print(f"{calls:8d} {total / 1e3:10.3f} {avg:10.2f} {pct:6.2f}  {name[:150]}")


if __name__ == "__main__":
    main()
