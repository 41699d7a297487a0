#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
KB="timeout 100 python tools/kbench.py"
V=$R/make-a-scene_amd/csrc/build/variants
for v in "" wg_noread wg_nomfma wg_nobar wg_nodma wg_noread_nodma wg_nomfma_nodma wg_noread_nodma_nobar; do
  echo "== ${v:-base}"
  for act in 0 2; do
    if [ -z "$v" ]; then $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; else MAS_HIP_LIB=$V/$v.so $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; fi
  done
done
