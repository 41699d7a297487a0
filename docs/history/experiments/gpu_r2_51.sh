#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_transformer.py tests/test_gpu_model.py -m gpu -q --timeout 600 2>&1 | tail -3
J='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["final_loss"])'
echo "== vq"; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
echo "== transformer"; timeout 600 python bench.py --workload transformer 2> /dev/null | python -c "$J"
echo "== e2e"; timeout 600 python bench.py --workload e2e 2> /dev/null | python -c "$J"
