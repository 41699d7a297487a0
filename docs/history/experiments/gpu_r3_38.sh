#!/bin/bash
# round-3 GPU call 48: Downsample forward on conv_s2_fwd_kernel vs conv_fwd.hip's stride-2 instance (MAS_CONV_S2=0 also switches the wgrad back)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
KB="timeout 120 python tools/kbench.py"
for v in 1 0; do
  echo "== [MAS_CONV_S2=$v]"
  MAS_CONV_S2=$v $KB conv_fwd --n 32 --c 128 --hw 256 --stride 2 --iters 30 2>&1 | tail -1
  MAS_CONV_S2=$v $KB conv_fwd --n 32 --c 128 --hw 128 --stride 2 --iters 30 2>&1 | tail -1
  MAS_CONV_S2=$v $KB conv_fwd --n 32 --c 256 --hw 64 --stride 2 --iters 30 2>&1 | tail -1
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0 1 0; do
  echo -n "bench [s2=$v]: "; MAS_CONV_S2=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
