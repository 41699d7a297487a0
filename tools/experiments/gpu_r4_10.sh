#!/bin/bash
# round 4, session 10: L2 behaviour of the spatial attention kernels (TCC hit / miss / request counters, fetch size)
cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_10; mkdir -p $O
{ bash tools/pmc_kernel.sh "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" sp_attn
  bash tools/pmc_kernel.sh "FETCH_SIZE TCP_TCC_READ_REQ_sum" sp_attn
  bash tools/pmc_kernel.sh "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" sp_attn; } 2>&1 | grep -v "^$" | cut -c1-400 > $O/pmc_sp.txt
cat $O/pmc_sp.txt
