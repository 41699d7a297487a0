#!/bin/bash
# round-2 GPU call 2: stream-kernel timeline (s_memtime), whole-epilogue ablation, SQ counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_2; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
{
for a in "0 0" "2 0" "2 1"; do MAS_HIP_LIB=$V/s_tl.so timeout 120 python tools/timeline_stream.py $a 2>&1 | grep -v amdgpu.ids; done
} | tee $O/timeline.txt
{
for act in 0 2; do
  echo -n "stream act=$act: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  for v in noepiall coreall; do echo -n "$v act=$act: "; MAS_HIP_LIB=$V/s_$v.so $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
for act in 0 2; do
 for cnt in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
   echo "== act=$act $cnt"
   bash tools/pmc_kernel.sh "$cnt" conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | grep -v amdgpu.ids | grep -i "stream\|error" | tail -3
 done
done 2>&1 | tee $O/pmc.txt
