#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
J='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'
for rep in 1 2; do
echo -n "fork on : "; timeout 600 python bench.py --workload transformer --steps 15 --warmup 8 2> /dev/null | python -c "$J"
echo -n "fork off: "; MAS_LN_FORK=0 timeout 600 python bench.py --workload transformer --steps 15 --warmup 8 2> /dev/null | python -c "$J"
done
