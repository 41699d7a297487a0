#!/usr/bin/env python3
"""Idle time inside one training step from a rocprofv3 --kernel-trace results.db: the window between the last two optimizer launches, the union
of busy intervals over all queues, and the largest gaps with the kernels on either side.  Usage: step_gaps.py results.db [top_n]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_multi" in r[2]]
    a, b = adam[-2] + 1, adam[-1] + 1
    win = rows[a:b]
    t0 = win[0][0]
    t1 = max(r[1] for r in win)
    cover_end, gaps, busy = win[0][0], [], 0
    last_name = "(previous step's adam_multi)"
    cur_start = win[0][0]
    for s, e, name in win:
        if s > cover_end:
            gaps.append((s - cover_end, last_name, name, cover_end - t0))
            busy += cover_end - cur_start
            cur_start = s
        if e > cover_end:
            cover_end, last_name = e, name
    busy += cover_end - cur_start
    short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "").replace("void ", "")[:44]
    tot_gap = sum(g[0] for g in gaps)
    print(f"step: {len(win)} launches, wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {tot_gap / 1e6:.3f} ms in {len(gaps)} gaps")
    hist = [(1, 0, 0), (2, 0, 0), (5, 0, 0), (10, 0, 0), (50, 0, 0), (1e9, 0, 0)]
    hist = [[h[0], 0, 0] for h in hist]
    for g in gaps:
        for h in hist:
            if g[0] / 1e3 < h[0]:
                h[1] += 1
                h[2] += g[0]
                break
    print("gap length histogram (us: count, total us): " + ", ".join(f"<{h[0]:g}: {h[1]}, {h[2] / 1e3:.0f}" for h in hist))
    for g in sorted(gaps, reverse=True)[:top]:
        print(f"  {g[0] / 1e3:8.1f} us at {g[3] / 1e6:7.3f} ms   after {short(g[1])}   before {short(g[2])}")


if __name__ == "__main__":
    main()
