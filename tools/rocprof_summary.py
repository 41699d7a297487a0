#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ms, % of GPU kernel time) from a rocprofv3
`--kernel-trace --stats` results.db (rocpd SQLite).  Usage: rocprof_summary.py results.db [out.txt]"""
import re
import sqlite3
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    dm = demangle([r[0] for r in rows])
    lines = [f"# rocprofv3 kernel-trace summary of {db}", f"# total GPU kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'%':>6} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel"]
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", dm.get(name, name))
        short = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", short)[:150]
        lines.append(f"{100*tot/total:6.2f} {n:7d} {tot/1e6:10.3f} {tot/n/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f}  {short}")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
