#!/bin/bash
# round-3 GPU call 21: what is the ~80 us floor of conv_wgrad_dma on the small maps?  (-DD_ABL_NOATOM: no split-K commit)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
for v in main wgrad_noatom; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
  echo "== [$v]"
  for s in "512 16" "512 32" "256 64" "128 128" "128 256"; do set -- $s
    env $L $KB wgrad --n 32 --c $1 --hw $2 --iters 30 2>&1 | tail -1
  done
done
echo "== trace of the small shape (kernel vs commit)"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pf_w -o w -- python $R/tools/kbench.py wgrad --n 32 --c 512 --hw 16 --iters 20 > /tmp/pf_w.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pf_w -name "*.db" | head -1) | head -8
