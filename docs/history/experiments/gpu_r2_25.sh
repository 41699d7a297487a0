#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
KB="timeout 100 python tools/kbench.py"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_wide.py -m gpu -q --timeout 600 2>&1 | tail -3
for s in "128 256" "128 128" "256 64" "512 32"; do set -- $s
  $KB gn_stats --n 32 --c $1 --hw $2 | tail -1; $KB gn_bwd --n 32 --c $1 --hw $2 | grep "^gn_bwd"
done
timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['encoder_stack']['fwd_ms'])"
