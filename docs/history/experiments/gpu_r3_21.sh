#!/bin/bash
# round-3 GPU call 25: GroupNorm backward on packed fp32 math (MAS_GN_BWD_PACKED=1) vs the generic kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_r3.py -m gpu -x -q 2>&1 | tail -3
KB="timeout 120 python tools/kbench.py"
for v in 1 0; do
  echo "== [MAS_GN_BWD_PACKED=$v]"
  for s in "128 256" "128 128" "256 64" "512 32"; do set -- $s
    MAS_GN_BWD_PACKED=$v $KB gn_bwd --n 32 --c $1 --hw $2 --iters 30 2>&1 | tail -1
  done
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0 1 0; do
  echo -n "bench [packed=$v]: "; MAS_GN_BWD_PACKED=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
MAS_GN_BWD_PACKED=$v timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pf_g$v -o g -- python $R/tools/kbench.py gn_bwd --n 32 --c 128 --hw 256 --iters 20 > /tmp/pf_g.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pf_g$v -name "*.db" | head -1) | head -7 | cut -c1-150
done
