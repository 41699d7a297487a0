#!/bin/bash
# round-3 GPU call 47: cache-policy bits of the thin forward kernel's output stores (aux 0 default / 1 sc0 / 2 nt / 3 both)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
for v in main thin_aux1 thin_aux2 thin_aux3 main; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
  echo -n "[$v] "; env $L $KB conv_fwd --n 32 --c 8 --co 128 --hw 256 --iters 50 2>&1 | tail -1
done
