#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 100 python tools/kbench.py"
for rep in 1 2; do
echo -n "plain act=0:   "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 0 | tail -1
echo -n "shipped act=2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 | tail -1
echo -n "no act math:   "; MAS_HIP_LIB=$V/wide_noactmath.so $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 | tail -1
echo -n "no act r/w:    "; MAS_HIP_LIB=$V/wide_noactrw.so $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 | tail -1
echo -n "shipped act=1: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 1 | tail -1
done
