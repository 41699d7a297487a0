#!/bin/bash
# round-3 GPU call 52: GroupNorm knobs re-checked in the step after the non-temporal stores (reverse walk, apply grid)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
run() { echo -n "bench [$1]: "; shift; env "$@" $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
run main X=1
run gn_reverse=0 MAS_GN_REVERSE=0
run apply_blocks=8192 MAS_GN_APPLY_BLOCKS=8192
run split_blocks=2048 MAS_GN_SPLIT_BLOCKS=2048
run main X=1
run gn_reverse=0 MAS_GN_REVERSE=0
