#!/bin/bash
# wgrad investigation evidence: ablations, padding experiments, row timeline, LDS issue-rate probe, PMC (-> gpurun_out/r2_wgrad/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r2_wgrad; mkdir -p $O
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 100 python tools/kbench.py"
{
for v in "" wg_noread wg_nomfma wg_nobar wg_nodma wg_noatom wg_noread_nodma_nobar wg_nomfma_nodma wg_pad_v10 wg_pad_v30 wg_pad_s30 wg_split2 ""; do
  echo "== ${v:-shipped}"
  for act in 0 2; do
    if [ -z "$v" ]; then $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; else MAS_HIP_LIB=$V/$v.so $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; fi
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/ablation.txt
{ MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 0; MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 2; } 2>&1 | grep -v amdgpu.ids > $O/timeline.txt
timeout 120 tools/probes/lds_rate > $O/lds_rate.txt 2>&1
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  echo "== $c"; bash tools/pmc_kernel.sh "$c" wgrad --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -2
done > $O/pmc.txt 2>&1
head -30 $O/ablation.txt; head -16 $O/timeline.txt; cat $O/pmc.txt | cut -c1-400
