#!/bin/bash
# round 4, session 27: SQ counters of the rewritten spatial attention kernels and of the causal attention kernels after the backward change
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_27; mkdir -p $O
{ echo "== spatial attention, 32 x 512 ch x 16^2 (tools/kbench.py sp_attn)"
  bash tools/pmc_kernel.sh "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" sp_attn
  bash tools/pmc_kernel.sh "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" sp_attn
  echo "== causal attention, B = 8 (tools/kbench.py attn --n 8)"
  bash tools/pmc_kernel.sh "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" attn --n 8
} 2>&1 | grep -v "^$" | cut -c1-330 > $O/pmc.txt; cat $O/pmc.txt
