#!/bin/bash
# HBM-side traffic of WHOLE training steps: rocprofv3 --pmc over `bench.py --steps K --warmup W`, one counter set per
# pass (own passes: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2; no trace domains besides --kernel-trace), every dispatch of the run.
# tools/step_traffic.py cuts the K timed steps out of the dispatch stream and sums per kernel family.
# usage (on the GPU box): bash tools/step_traffic.sh <outdir> [K] [W]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$1; K=${2:-3}; W=${3:-2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC_EA0_[A-Z0-9_]*(RDREQ|WRREQ|RD_|WR_)[A-Z0-9_]*|FETCH_SIZE|WRITE_SIZE|TCC_EA0_[A-Z0-9_]*DRAM[A-Z0-9_]*|TCC_[A-Z0-9_]*MALL[A-Z0-9_]*)\b" | sort -u > $R/$O/counters_available.txt
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1)); rm -rf /tmp/st_$i
  timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/st_$i -o st --output-format csv -- python $R/bench.py --steps $K --warmup $W --no-cpu-baseline --no-also --no-encoder-stack > /tmp/st_$i.log 2>&1 || { echo "pass $i ($CNT) failed"; tail -5 /tmp/st_$i.log; continue; }
  f=$(find /tmp/st_$i -name "*counter_collection.csv" | head -1)
  # keep only what the summariser needs: dispatch id, kernel name, counter name, value (the raw csv is ~100 MB)
  python3 - "$f" "$R/$O/pass_$i.csv" <<'PY'
import csv, sys
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
for r in csv.DictReader(open(sys.argv[1])):
    w.writerow([r["Dispatch_Id"], r["Kernel_Name"][:90], r["Counter_Name"], r["Counter_Value"]])
PY
  grep -o '"ms_per_step": [0-9.]*' /tmp/st_$i.log | head -1 | sed "s/^/pass $i ($CNT) under profiling: /"
done
cd $R && python3 tools/step_traffic.py $O $K $W | tee $O/summary.txt
