#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -4
J='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["final_loss"])'
echo "== vq"; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
echo "== vq again"; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
