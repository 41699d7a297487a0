#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_transformer.py tests/test_gpu_sampling.py tests/test_gpu_dp.py -m gpu -q --timeout 600 2>&1 | tail -3
echo "== transformer"; timeout 600 python bench.py --workload transformer 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['final_loss'])"
echo "== e2e"; timeout 600 python bench.py --workload e2e 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
