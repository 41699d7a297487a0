#!/bin/bash
# round-3 GPU call 27: GroupNorm backward: non-temporal dx stores x grid size; gn_act with a non-temporal store; step A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
run() { echo "== [$1]"; shift; for s in "128 256" "128 128" "256 64" "512 32"; do env "$@" $KB gn_bwd --n 32 --c ${s% *} --hw ${s#* } --iters 30 2>&1 | tail -1; done; }
run main X=1
run nts MAS_HIP_LIB=$V/gn_nts.so
run nts+apply4096 MAS_HIP_LIB=$V/gn_nts.so MAS_GN_APPLY_BLOCKS=4096
run nts+apply8192 MAS_HIP_LIB=$V/gn_nts.so MAS_GN_APPLY_BLOCKS=8192
run nts+apply16384 MAS_HIP_LIB=$V/gn_nts.so MAS_GN_APPLY_BLOCKS=16384
run nts+apply4096+split2048 MAS_HIP_LIB=$V/gn_nts.so MAS_GN_APPLY_BLOCKS=4096 MAS_GN_SPLIT_BLOCKS=2048
for v in main gn_actnts; do
if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
echo "== gn_act [$v]"
for s in "128 256" "128 128" "256 64"; do env $L $KB gn_act --n 32 --c ${s% *} --hw ${s#* } --iters 30 2>&1 | tail -1; done
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
bench() { echo -n "bench [$1]: "; shift; env "$@" $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
bench main X=1
bench nts+4096 MAS_HIP_LIB=$V/gn_nts.so MAS_GN_APPLY_BLOCKS=4096
bench actnts+4096 MAS_HIP_LIB=$V/gn_actnts.so MAS_GN_APPLY_BLOCKS=4096
bench main X=1
bench nts+4096 MAS_HIP_LIB=$V/gn_nts.so MAS_GN_APPLY_BLOCKS=4096
bench actnts+4096 MAS_HIP_LIB=$V/gn_actnts.so MAS_GN_APPLY_BLOCKS=4096
