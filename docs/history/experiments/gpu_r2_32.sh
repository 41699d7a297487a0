#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 100 python tools/kbench.py"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout 600 2>&1 | tail -4
for s in "128 256" "128 128" "256 128" "256 64" "512 32"; do set -- $s
  for act in 0 2; do $KB wgrad --n 32 --c $1 --hw $2 --act $act | tail -1; done
done
MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 0 2>&1 | grep -v amdgpu.ids | head -14
MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 2 2>&1 | grep -v amdgpu.ids | head -14
