#!/bin/bash
# Per-kernel PMC sums for one kbench invocation.  usage: tools/pmc_kernel.sh "<counters>" <kbench args...>
# (counters in their own pass, no trace domains: gpurun refuses --pmc combined with sys/hip traces)
set -e
CNT="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_out
timeout 200 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/pmc_out -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/kbench.py "$@" --iters 3 > /tmp/pmc.log 2>&1 || { tail -5 /tmp/pmc.log; exit 1; }
f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
k=$(find /tmp/pmc_out -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$k" <<'PY'
import csv, sys, collections
try:      # per-kernel average duration of the SAME profiled run (clock = cycles / duration)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k_, v in dur.items():
        if "conv" in k_ or "gn_" in k_ or "attn" in k_ or "vq" in k_:
            print(k_, "avg duration us under profiling", round(sum(v) / len(v) / 1e3, 1), "n", len(v))
except Exception as e:
    print("no kernel trace:", e)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k, d in acc.items():
    if "conv" in k or "gn_" in k or "attn" in k or "vq" in k:
        print(k, "dispatches", cnt[k], {c: v / cnt[k] for c, v in d.items()})
PY
