#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
KB="timeout 100 python tools/kbench.py"
V=$R/make-a-scene_amd/csrc/build/variants
for v in "" wg_pad_v10 wg_pad_v30 wg_pad_s30 ""; do
  echo "== ${v:-base}"
  if [ -z "$v" ]; then $KB wgrad --n 32 --c 128 --hw 256 --act 0 | tail -1; else MAS_HIP_LIB=$V/$v.so $KB wgrad --n 32 --c 128 --hw 256 --act 0 | tail -1; fi
done
