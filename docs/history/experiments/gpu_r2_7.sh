#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_7; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
{
echo -n "wide: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -1
for v in fake nt sc1 sc0sc1 noepi; do echo -n "$v: "; MAS_HIP_LIB=$V/w_$v.so $KB conv_fwd --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -1; done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
