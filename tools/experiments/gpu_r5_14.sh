#!/bin/bash
# round 5, call 14: does the causal attention forward re-read K / V from beyond the L2?  rocprofv3 --pmc (own passes) on kbench attn
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_14; mkdir -p $O
{
for k in "attn --n 8 --iters 20"; do
  bash tools/pmc_kernel.sh "FETCH_SIZE" $k
  bash tools/pmc_kernel.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $k
  bash tools/pmc_kernel.sh "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $k
done
} > $O/attn_pmc.txt 2>&1; grep -v "avg duration" $O/attn_pmc.txt | cut -c1-260
