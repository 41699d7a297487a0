#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_r2.py -m gpu -q --timeout 600 2>&1 | tail -3
J='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'
for rep in 1 2; do
echo -n "ticket on : "; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
echo -n "ticket off: "; MAS_GN_TICKET=0 timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "$J"
done
timeout 100 python tools/kbench.py gn_stats --n 32 --c 128 --hw 256 | tail -1
timeout 100 python tools/kbench.py gn_bwd --n 32 --c 128 --hw 256 | grep "^gn_bwd"
MAS_GN_TICKET=0 timeout 100 python tools/kbench.py gn_stats --n 32 --c 128 --hw 256 | tail -1
MAS_GN_TICKET=0 timeout 100 python tools/kbench.py gn_bwd --n 32 --c 128 --hw 256 | grep "^gn_bwd"
