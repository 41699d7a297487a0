#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py -m gpu -q --timeout 600 2>&1 | grep -B2 -A12 "Error\|^E " | head -60
