#!/bin/bash
# round 5, call 24: the causal attention kernels' counters on the final build (VERDICT r4 next #4 asked for r05_attn_pmc.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_24; mkdir -p $O
{
echo "# round 5, final build: rocprofv3 --pmc (three separate passes, tools/pmc_kernel.sh) on tools/kbench.py attn --n 8 (B=8, H=16, S=1536, hd=64, bf16)"
bash tools/pmc_kernel.sh "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" attn --n 8
bash tools/pmc_kernel.sh "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" attn --n 8
bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" attn --n 8
} > $O/attn_pmc.txt 2>&1; cut -c1-300 $O/attn_pmc.txt
