#!/bin/bash
# round-3 GPU call 26: what bounds the GroupNorm backward passes at ~5 TB/s?  non-temporal access, loads in flight, grid size
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
run() { echo "== [$1]"; shift; for s in "128 256" "128 128" "256 64"; do set -- "$@"; env "$@" $KB gn_bwd --n 32 --c ${s% *} --hw ${s#* } --iters 30 2>&1 | tail -1; done; }
run main X=1
run ntl MAS_HIP_LIB=$V/gn_ntl.so
run ntls MAS_HIP_LIB=$V/gn_ntls.so
run unr4 MAS_HIP_LIB=$V/gn_unr4.so
run ntls4 MAS_HIP_LIB=$V/gn_ntls4.so
run split2048 MAS_GN_SPLIT_BLOCKS=2048
run split512 MAS_GN_SPLIT_BLOCKS=512
run apply4096 MAS_GN_APPLY_BLOCKS=4096
run apply1024 MAS_GN_APPLY_BLOCKS=1024
run main X=1
cd /tmp && export TMPDIR=/tmp
for v in main gn_ntls gn_ntls4; do
if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
env $L timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pf_$v -o g -- python $R/tools/kbench.py gn_bwd --n 32 --c 128 --hw 256 --iters 20 > /tmp/pf_g.log 2>&1
echo "trace [$v]"; python $R/tools/rocprof_summary.py $(find /tmp/pf_$v -name "*.db" | head -1) | sed -n 4,5p | cut -c1-110
done
