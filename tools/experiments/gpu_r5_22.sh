#!/bin/bash
# round 5, call 22: counters behind DESIGN section 3 #7 -- how busy is the VALU in the GroupNorm backward passes, and what do they fetch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_22; mkdir -p $O
{
echo "# rocprofv3 --pmc (own passes, tools/pmc_kernel.sh) on tools/kbench.py gn_bwd --n 32 --c 128 --hw 256 --three 1 (gn_bwd_partial_pk<true>, gn_bwd_apply_pk<true,false>) and gn_act"
for k in "gn_bwd --n 32 --c 128 --hw 256 --three 1" "gn_act --n 32 --c 128 --hw 256"; do
  echo "== kbench $k"
  bash tools/pmc_kernel.sh "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" $k
  bash tools/pmc_kernel.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $k
  bash tools/pmc_kernel.sh "FETCH_SIZE" $k
  bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE WRITE_SIZE" $k
done
} > $O/gn_pmc.txt 2>&1; cut -c1-330 $O/gn_pmc.txt
