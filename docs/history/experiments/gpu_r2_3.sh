#!/bin/bash
# round-2 GPU call 3: late issue of the HBM-facing operations / static priority for the second-dispatched half
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_3; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
for v in late lateprio; do
  echo "== stream_check $v"; MAS_HIP_LIB=$V/s_$v.so MAS_CONV_STREAM_MIN_TILES_PER_CU=0 timeout 300 python tests/helpers/stream_check.py 2>&1 | grep -v amdgpu.ids | grep -c "^ok"
done
{
for act in 0 2; do
  echo -n "base act=$act: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  for v in late prio lateprio; do echo -n "$v act=$act: "; MAS_HIP_LIB=$V/s_$v.so $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1; done
done
echo -n "base res: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 2>&1 | tail -1
echo -n "late res: "; MAS_HIP_LIB=$V/s_late.so $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
MAS_HIP_LIB=$V/s_late_tl.so timeout 120 python tools/timeline_stream.py 0 0 2>&1 | grep -v amdgpu.ids | head -14 | tee $O/timeline_late.txt
