#!/bin/bash
# round-3 GPU call 35: 1x1 convolutions on the GEMM kernel (conv1x1.hip) vs conv_fwd.hip's KS = 1 instance (MAS_CONV1X1=0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_spatial_attn.py -m gpu -x -q 2>&1 | tail -4
KB="timeout 120 python tools/kbench.py"
for v in 1 0; do
  echo "== [MAS_CONV1X1=$v]"
  for s in "512 1536 16" "512 512 16" "1536 512 16" "256 128 128" "128 256 128" "512 256 64" "256 512 64"; do set -- $s
    MAS_CONV1X1=$v $KB conv_fwd --n 32 --c $1 --co $2 --hw $3 --ks 1 --iters 50 2>&1 | tail -1
  done
  MAS_CONV1X1=$v $KB conv_fwd --n 32 --c 512 --co 512 --hw 16 --ks 1 --res 1 --iters 50 2>&1 | tail -1
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0 1 0; do
  echo -n "bench [conv1x1=$v]: "; MAS_CONV1X1=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
