#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -3
J='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["final_loss"])'
echo "== vq"; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R; python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) /tmp/kt.txt > /dev/null; grep "pack_weight" /tmp/kt.txt | cut -c1-120
