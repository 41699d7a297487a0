#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_12; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_losses.py -m gpu -q -x -rP --timeout 600 > $O/pytest_losses.txt 2>&1; tail -30 $O/pytest_losses.txt | cut -c1-220
echo "== full"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4
