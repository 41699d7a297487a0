#!/bin/bash
# round-3 GPU call 22: split-K partials + fixed-order reduce (no fp32 atomics) vs the atomic commit: tests, kbench, step A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -8
KB="timeout 120 python tools/kbench.py"
for v in 1 0; do
  echo "== [MAS_WGRAD_PARTIALS=$v]"
  for s in "512 16" "512 32" "256 64" "128 128" "128 256"; do set -- $s
    MAS_WGRAD_PARTIALS=$v $KB wgrad --n 32 --c $1 --hw $2 --iters 30 2>&1 | tail -1
  done
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0; do
  echo -n "bench [partials=$v]: "; MAS_WGRAD_PARTIALS=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
echo "== trace of the small shape"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pf_w -o w -- python $R/tools/kbench.py wgrad --n 32 --c 512 --hw 16 --iters 20 > /tmp/pf_w.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pf_w -name "*.db" | head -1) | head -8
