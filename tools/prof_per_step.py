#!/usr/bin/env python
"""Per-STEP kernel counts and times of bench.py, free of set-up work: two rocprofv3 --kernel-trace --stats runs of the same command that
differ only in --steps; (B - A) / (steps_B - steps_A) per kernel.
    python tools/prof_per_step.py A_results.db stepsA B_results.db stepsB > profiles/rNN_per_step.txt"""
import re
import sqlite3
import sys


def load(path):
    cur = sqlite3.connect(path).cursor()
    return {n: (c, t) for n, c, t in cur.execute("select name, total_calls, total_duration from top_kernels")}


def main():
    a, sa, b, sb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    ds = sb - sa
    rows = []
    for name in set(a) | set(b):
        ca, ta = a.get(name, (0, 0.0))
        cb, tb = b.get(name, (0, 0.0))
        rows.append(((tb - ta) / ds, (cb - ca) / ds, name))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"# per bench step = ({sys.argv[3]} - {sys.argv[1]}) / {ds} steps: launches per step, GPU time per step (us), share")
    print(f"# GPU busy time per step {tot / 1e3:.3f} ms, {sum(r[1] for r in rows):.1f} launches per step")
    print(f"{'launches':>9} {'us/step':>10} {'avg_us':>8} {'pct':>6}  kernel")
    for us, n, name in rows[:45]:
        if abs(n) < 1e-9 and abs(us) < 1e-3:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"unsigned short", "bf16", name)
        print(f"{n:9.1f} {us:10.1f} {us / n if n else 0:8.2f} {100 * us / tot if tot else 0:6.2f}  {name[:140]}")


if __name__ == "__main__":
    main()
