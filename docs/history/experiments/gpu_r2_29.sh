#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
KB="timeout 100 python tools/kbench.py"
V=$R/make-a-scene_amd/csrc/build/variants
for v in "" wg_noatom wg_noatom_noread_nodma_nobar wg_noread_nodma_nobar; do
  echo "== ${v:-base}"
  for n in 32 8; do
    if [ -z "$v" ]; then $KB wgrad --n $n --c 128 --hw 256 --act 0 | tail -1; else MAS_HIP_LIB=$V/$v.so $KB wgrad --n $n --c 128 --hw 256 --act 0 | tail -1; fi
  done
done
