#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_13; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_losses.py -m gpu -q -rP --timeout 600 > $O/pytest_losses.txt 2>&1; grep -h "passed\|failed\|rel-L2\|disc logits\|^E " $O/pytest_losses.txt | cut -c1-200
