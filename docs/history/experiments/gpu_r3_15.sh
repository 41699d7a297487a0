#!/bin/bash
# round-3 GPU call 18: evidence for the transformer side: kernel trace of the MakeAScene step, PMC of the attention kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_15; mkdir -p $O
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_tr -o tr -- python $R/bench.py --workload transformer --steps 4 --warmup 2 > /tmp/pf_tr.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_tr -name "*.db" | head -1) $O/kernel_trace_transformer.txt > /dev/null 2>&1; head -28 $O/kernel_trace_transformer.txt
echo "== PMC attention (B=8: fwd v2, then fwd+bwd)"
bash tools/pmc_kernel.sh "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" attn --n 8 2>&1 | grep -v amdgpu.ids | tee $O/pmc_attn1.txt
bash tools/pmc_kernel.sh "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" attn --n 8 2>&1 | grep -v amdgpu.ids | tee $O/pmc_attn2.txt
bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" attn --n 8 2>&1 | grep -v amdgpu.ids | tee $O/pmc_attn3.txt
