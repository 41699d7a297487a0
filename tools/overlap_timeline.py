#!/usr/bin/env python3
"""Timeline of one ResnetBlock backward from a rocprofv3 --kernel-trace results.db (MAS_OVERLAP experiments): around the LAST
537 MB-class gn_bwd_apply launch of the trace, prints every kernel with start / end relative to the window, its queue id and
duration -- to see whether the weight gradient and the GroupNorm backward really ran side by side and what the switches cost.
Also the step summary (wall between the last two optimizer bursts, union of busy time).  Usage: overlap_timeline.py results.db"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute(f"select d.start, d.end, s.kernel_name, d.{q} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    adam = [i for i, r in enumerate(rows) if "FusedAdam" in r[2] or "adam_multi" in r[2]]
    ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
    a, b = ends[-2] + 1, ends[-1] + 1
    win = rows[a:b]
    t0, t1 = win[0][0], max(r[1] for r in win)
    ev = sorted([(r[0], 1) for r in win] + [(r[1], -1) for r in win])
    busy, depth, last = 0, 0, t0
    for t, d in ev:
        if depth > 0:
            busy += t - last
        depth += d
        last = t
    print(f"step: {len(win)} launches, wall {(t1 - t0) / 1e6:.3f} ms, sum of kernel durations {sum(r[1] - r[0] for r in win) / 1e6:.3f} ms, "
          f"union busy {busy / 1e6:.3f} ms ({100 * busy / (t1 - t0):.1f} %)")
    big = [i for i, r in enumerate(win) if "gn_bwd_apply" in r[2] and r[1] - r[0] > 300e3]
    if not big:
        print("no large gn_bwd_apply in the window"); return
    c = big[len(big) // 2]
    lo, hi = max(0, c - 14), min(len(win), c + 10)
    base = win[lo][0]
    for s, e, name, qid in win[lo:hi]:
        short = name.replace("_ZN12_GLOBAL__N_1", "").replace("void ", "")[:46]
        print(f"  {(s - base) / 1e3:9.1f} -> {(e - base) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{qid}  {short}")


if __name__ == "__main__":
    main()
