#!/usr/bin/env python3
"""GPU busy fraction of one training step from a rocprofv3 --kernel-trace results.db: the window between the last two optimizer
launches (multi_tensor_apply ... FusedAdam) -- wall time, sum of kernel durations, number of launches, idle gaps by size, and the
kernels that precede the largest gaps.  Usage: step_gaps.py results.db"""
import sqlite3
import sys
from collections import Counter


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    adam = [i for i, r in enumerate(rows) if "FusedAdam" in r[2] or "multi_tensor_apply" in r[2]]
    # group consecutive optimizer launches into steps
    ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
    if len(ends) < 2:
        print("need two optimizer steps in the trace"); return
    a, b = ends[-2] + 1, ends[-1] + 1
    win = rows[a:b]
    wall = win[-1][1] - win[0][0]
    busy = sum(e - s for s, e, _ in win)
    gaps = [(win[i + 1][0] - win[i][1], win[i][2], win[i + 1][2]) for i in range(len(win) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"step window: {len(win)} launches, wall {wall / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms ({100 * busy / wall:.1f} % busy), "
          f"idle {sum(g[0] for g in pos) / 1e6:.3f} ms in {len(pos)} gaps (overlap {-sum(g[0] for g in gaps if g[0] < 0) / 1e6:.3f} ms)")
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)):
        sel = [g[0] for g in pos if lo * 1e3 <= g[0] < hi * 1e3]
        print(f"  gaps {lo:>3}-{hi if hi < 1e9 else 'inf':>3} us: {len(sel):5d}  total {sum(sel) / 1e6:.3f} ms")
    c = Counter()
    for g, before, after in pos:
        c[(before[:50], after[:50])] += g
    print("largest idle by (kernel before -> kernel after):")
    for (bk, ak), t in c.most_common(12):
        print(f"  {t / 1e6:7.3f} ms  {bk}  ->  {ak}")


if __name__ == "__main__":
    main()
