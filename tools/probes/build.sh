#!/bin/bash
# Builds the micro-probes next to their sources (binaries are git-ignored; they travel to the GPU box with the gpurun snapshot).
cd "$(dirname "$0")"
for p in coissue mfma_peak mfma_power fetch_calib lds_rate tr_probe dma_oob_probe dma_imm_probe dma_hi_probe cu_stream wino_loop; do
  [ -f $p.hip ] && hipcc --offload-arch=gfx950 -O3 -o $p $p.hip 2>/dev/null && echo built $p
done
