#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 100 python tools/kbench.py"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout 600 2>&1 | tail -2
for s in "128 256" "128 128" "256 128" "256 64" "512 32"; do set -- $s
  for act in 0 2; do $KB wgrad --n 32 --c $1 --hw $2 --act $act | tail -1; done
done
echo "== split2 (spilling)"
for act in 0 2; do MAS_HIP_LIB=$V/wg_split2.so $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
MAS_HIP_LIB=$V/wg_split2.so timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -k wgrad 2>&1 | tail -2
