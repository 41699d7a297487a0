#!/bin/bash
# round-3 GPU call 44: wide kernel for shapes with 1 tile per CU (MAS_CONV_WIDE_MIN_TILES_PER_CU=1) vs 2 (default)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
KB="timeout 120 python tools/kbench.py"
for v in 2 1; do
  echo "== [MAS_CONV_WIDE_MIN_TILES_PER_CU=$v]"
  MAS_CONV_WIDE_MIN_TILES_PER_CU=$v $KB conv_fwd --n 32 --c 512 --hw 32 --iters 50 2>&1 | tail -1
  MAS_CONV_WIDE_MIN_TILES_PER_CU=$v $KB dgrad --n 32 --c 512 --hw 32 --iters 50 2>&1 | tail -1
  MAS_CONV_WIDE_MIN_TILES_PER_CU=$v $KB conv_fwd --n 32 --c 256 --co 512 --hw 32 --iters 50 2>&1 | tail -1
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 2 1 2 1; do
  echo -n "bench [wide_min=$v]: "; MAS_CONV_WIDE_MIN_TILES_PER_CU=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
