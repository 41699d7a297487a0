#!/bin/bash
# round-3 GPU call 53: what would fewer LDS bytes per MFMA buy the wide kernel?  -DW_ABL_HALF_A: 0.5 instead of 0.75 fragment reads per MFMA (wrong results)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
for v in main wide_halfa main wide_halfa; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
  echo -n "[$v] "; env $L $KB conv_fwd --n 32 --c 128 --hw 256 --iters 30 2>&1 | tail -1
  echo -n "[$v] "; env $L $KB conv_fwd --n 32 --c 512 --hw 64 --iters 30 2>&1 | tail -1
done
