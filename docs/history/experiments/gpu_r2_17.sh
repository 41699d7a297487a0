#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_17; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
{
for act in 0 2; do
  echo -n "dma act=$act: "; $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1
  for v in noatom nodma core; do echo -n "$v act=$act: "; MAS_HIP_LIB=$V/d_$v.so $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
for cnt in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
   echo "== $cnt"
   bash tools/pmc_kernel.sh "$cnt" wgrad --n 32 --c 128 --hw 256 --act 0 2>&1 | grep -v amdgpu.ids | grep -i "wgrad\|error" | tail -2
done 2>&1 | tee $O/pmc.txt
