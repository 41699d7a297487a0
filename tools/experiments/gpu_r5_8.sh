#!/bin/bash
# round 5, call 8: smoke() with the sub-pixel kernels; rocprofv3 --pmc (own passes) on the new kernels: HBM traffic against the algorithmic bytes, matrix-pipe busy
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_8; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "Warning\|warn\|rel = lambda\|amdgpu.ids" | tail -6
{
echo "# rocprofv3 --pmc (separate passes, tools/pmc_kernel.sh) on tools/kbench.py: Upsample conv 128->128 @128->256, B = 32 -- conv_up2_kernel<true> (forward + statistics) and conv_wgrad_dma_kernel<false,1,2> (weight gradient)"
for k in "conv_fwd --c 128 --hw 128 --ups 1 --stats 1" "wgrad --c 128 --hw 128 --ups 1"; do
  echo "== kbench $k"
  bash tools/pmc_kernel.sh "FETCH_SIZE" $k
  bash tools/pmc_kernel.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $k
  bash tools/pmc_kernel.sh "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" $k
  bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $k
done
} > $O/up2_pmc.txt 2>&1; grep -v "pack_\|avg duration" $O/up2_pmc.txt | cut -c1-300
