#!/bin/bash
# round-3 GPU call 36: 1x1 GEMM kernel + materialised GroupNorm in front of it (q / k / v), step A/B and the attention-block tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_spatial_attn.py tests/test_gpu_parity_r2.py -m gpu -x -q 2>&1 | tail -2
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0 1 0; do
  echo -n "bench [conv1x1=$v]: "; MAS_CONV1X1=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
timeout 250 python tools/conv_shape_profile.py 2>&1 | grep -E " 1 1 |^step|^kind"
