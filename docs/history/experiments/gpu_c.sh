#!/bin/bash
# round-2 GPU call C: old-vs-new MFMA core, SQ counters of the stream kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
{
echo -n "old core: "; MAS_CONV_STREAM=0 MAS_HIP_LIB=$V/oldcore.so timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1
echo -n "new core: "; MAS_HIP_LIB=$V/s_core.so timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1
echo -n "old full: "; MAS_CONV_STREAM=0 timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1
echo -n "new full: "; timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1
} 2>&1 | tee $O/core.txt
for lib in "" "$V/s_core.so"; do
 for cnt in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
   echo "== lib=${lib:-shipped} $cnt"
   if [ -n "$lib" ]; then export MAS_HIP_LIB=$lib; else unset MAS_HIP_LIB; fi; bash tools/pmc_kernel.sh "$cnt" conv_fwd --n 32 --c 128 --hw 256 2>&1 | grep -v amdgpu.ids | tail -3
 done
done 2>&1 | tee $O/pmc.txt
