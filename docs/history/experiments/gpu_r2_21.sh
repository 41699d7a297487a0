#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_model.py -m gpu -q --timeout 600 2>&1 | tail -3
for i in 1 2; do
echo -n "fused on : "; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['populations'], j['encoder_stack']['fwd_ms'])"
echo -n "fused off: "; MAS_FUSED_GN_STATS=0 timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['populations'], j['encoder_stack']['fwd_ms'])"
done
