#!/bin/bash
# round-3 GPU call 45: full GPU suite on the current state + wide-kernel eligibility A/B (min tiles per CU 1 vs 0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0 1 0; do
  echo -n "bench [wide_min=$v]: "; MAS_CONV_WIDE_MIN_TILES_PER_CU=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
