#!/bin/bash
# Per-kernel PMC sums for one kbench invocation.  usage: tools/pmc_kernel.sh "<counters>" <kbench args...>
# (counters in their own pass, no trace domains: gpurun refuses --pmc combined with sys/hip traces)
set -e
CNT="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_out
timeout 200 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/pmc_out -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/kbench.py "$@" --iters 3 > /tmp/pmc.log 2>&1 || { tail -5 /tmp/pmc.log; exit 1; }
f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
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
