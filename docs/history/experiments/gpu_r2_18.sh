#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_18; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
{
for act in 0 2; do
  echo -n "dma act=$act: "; $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1
  echo -n "spread act=$act: "; MAS_HIP_LIB=$V/d_spread.so $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1
done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
MAS_HIP_LIB=$V/d_spread.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_r2.py -m gpu -q -x --timeout 600 2>&1 | tail -2
