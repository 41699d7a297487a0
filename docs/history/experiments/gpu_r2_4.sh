#!/bin/bash
# round-2 GPU call 4: first run of the wide 3x3 kernel: correctness, kbench vs the stream kernel, timeline
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_4; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
echo "== wide_check"; MAS_CONV_WIDE_MIN_TILES_PER_CU=0 timeout 300 python tests/helpers/wide_check.py 2>&1 | grep -v amdgpu.ids | tail -16 | tee $O/wide_check.txt
{
for act in 0 2; do
  echo -n "stream act=$act: "; MAS_CONV_WIDE=0 $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  echo -n "wide   act=$act: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
done
echo -n "wide res act=2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 2>&1 | tail -1
echo -n "wide dgrad: "; $KB dgrad --n 32 --c 128 --hw 256 2>&1 | tail -1
for s in "256 64" "512 32" "128 128" "256 128" "512 64"; do set -- $s
  echo -n "c$1 hw$2 wide: "; MAS_CONV_WIDE_MIN_TILES_PER_CU=1 $KB conv_fwd --n 32 --c $1 --hw $2 --act 2 2>&1 | tail -1
  echo -n "c$1 hw$2 stream: "; MAS_CONV_WIDE=0 $KB conv_fwd --n 32 --c $1 --hw $2 --act 2 2>&1 | tail -1
done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
for a in "0 0" "2 0"; do TL_KERNEL=wide MAS_HIP_LIB=$V/w_tl.so timeout 120 python tools/timeline_stream.py $a 2>&1 | grep -v amdgpu.ids | head -12; done | tee $O/timeline_wide.txt
