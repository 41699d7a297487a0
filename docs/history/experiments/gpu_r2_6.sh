#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_6; mkdir -p $O
cd $R
KB="timeout 120 python tools/kbench.py"
{
for st in 0 1 2; do
  echo -n "wide stagger=$st act=0: "; MAS_CONV_WIDE_STAGGER=$st $KB conv_fwd --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -1
done
for w in 1 2 4 8; do echo -n "wgs/cu=$w: "; MAS_CONV_WGS_PER_CU=$w $KB conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1; done
echo -n "n=64: "; $KB conv_fwd --n 64 --c 128 --hw 256 2>&1 | tail -1
echo -n "n=64 stagger1: "; MAS_CONV_WIDE_STAGGER=1 $KB conv_fwd --n 64 --c 128 --hw 256 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
