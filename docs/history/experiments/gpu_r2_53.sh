#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
J='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["final_loss"])'
for rep in 1 2; do
echo -n "batched  : "; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
echo -n "per image: "; MAS_PACK_BATCH=0 timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
done
