#!/bin/bash
# round-3 GPU call 42: Downsample weight gradient on wgrad_s2_kernel (MAS_CONV_S2=1) vs conv_wgrad_kernel<3,2,4>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0 1 0; do
  echo -n "bench [s2=$v]: "; MAS_CONV_S2=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
timeout 250 python tools/conv_shape_profile.py 2>&1 | grep -E " 3 2 |^step|^kind"
MAS_CONV_S2=0 timeout 250 python tools/conv_shape_profile.py 2>&1 | grep -E "wgrad.* 3 2 "
