#!/bin/bash
# round 6, call 12: the request-size counters on tools/probes/fetch_calib (a full-line stream and a half-line stream of known byte counts):
# does a 64-byte-of-every-128 stream show up as 64-byte or as 128-byte fabric requests?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_12; mkdir -p $O
[ -x tools/probes/fetch_calib ] || bash tools/probes/build.sh > /dev/null 2>&1
cd /tmp
for CNT in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/fc; timeout 120 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/fc -o fc --output-format csv -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib > /tmp/fc.log 2>&1
  grep "requests" /tmp/fc.log | head -1
  python3 - "$(find /tmp/fc -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Dispatch_Id"], r["Kernel_Name"][:10], r["Counter_Name"])
    acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
for (d, n, c), v in acc.items():
    if n.startswith("k_"): print(f"  dispatch {d} {n} {c} = {v:.6g}")
PY
done 2>&1 | tee $O/fetch_calib_request_sizes.txt
