#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_19; mkdir -p $O
cd $R
KB="timeout 120 python tools/kbench.py"
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_transformer.py -m gpu -q -x --timeout 600 2>&1 | tail -3
echo "== full"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4
echo "== bench (fused stats on / off)"
timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | cut -c1-330
MAS_FUSED_GN_STATS=0 timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | cut -c1-330
echo "== transformer"
timeout 600 python bench.py --workload transformer --steps 5 --warmup 2 2> /dev/null | cut -c1-300
{ echo -n "wide plain: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -1; echo -n "wide act2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 2>&1 | tail -1; } | grep -v amdgpu
