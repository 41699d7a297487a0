#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 100 python tools/kbench.py"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "wgrad or conv" 2>&1 | tail -2
for v in "" wg_prio; do
  echo "== ${v:-base}"
  for act in 0 2; do
    if [ -z "$v" ]; then $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; else MAS_HIP_LIB=$V/$v.so $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; fi
  done
done
MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 0 2>&1 | grep -v amdgpu.ids | head -14
MAS_HIP_LIB=$V/wg_tl_prio.so timeout 120 python tools/timeline_wgrad.py 0 2>&1 | grep -v amdgpu.ids | head -14
MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 2 2>&1 | grep -v amdgpu.ids | head -14
