#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_14; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_spatial_attn.py -m gpu -q -rP --timeout 600 > $O/pytest_sp.txt 2>&1; grep -h "passed\|failed\|rel-L2\|^E " $O/pytest_sp.txt | cut -c1-200
echo "== full"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | cut -c1-330
