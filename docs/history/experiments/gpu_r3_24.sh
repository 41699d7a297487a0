#!/bin/bash
# round-3 GPU call 28: causal attention grid in longest-processing-time-first order (MAS_ATTN_LPT)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_transformer.py tests/test_gpu_kernels.py -m gpu -x -q -k "attn or attention or transformer or scene" 2>&1 | tail -3
KB="timeout 120 python tools/kbench.py"
for v in 1 0 1 0; do
  echo "== [MAS_ATTN_LPT=$v]"
  MAS_ATTN_LPT=$v $KB attn --n 8 --iters 50 2>&1 | tail -2
done
for v in 1 0; do
  echo "== [MAS_ATTN_LPT=$v] B=1 / B=32"
  MAS_ATTN_LPT=$v $KB attn --n 1 --iters 50 2>&1 | tail -2
  MAS_ATTN_LPT=$v $KB attn --n 32 --iters 30 2>&1 | tail -2
done
for v in 1 0; do
echo -n "transformer step [lpt=$v]: "; MAS_ATTN_LPT=$v timeout 300 python bench.py --workload transformer --no-cpu-baseline --no-also --steps 12 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %s  %.3f ms/step' % (d['value'], d['unit'], d['ms_per_step']))"
done
