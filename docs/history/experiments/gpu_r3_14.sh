#!/bin/bash
# round-3 GPU call 17: wgrad with kw-shifted fragment reuse (50 instead of 80 transpose reads per tile): parity + A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_14; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
echo "== pytest (conv kernels, parity r2/r3, model)"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_model.py tests/test_gpu_losses.py -m gpu -q 2>&1 | tail -3
KB="timeout 120 python tools/kbench.py"
for v in main nokw main nokw; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/wgrad_$v.so"; fi
  echo -n "kbench wgrad [$v]: "; env $L $KB wgrad --n 32 --c 128 --hw 256 2>&1 | tail -1
  echo -n "kbench wgrad c256 hw64 [$v]: "; env $L $KB wgrad --n 32 --c 256 --hw 64 2>&1 | tail -1
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in main nokw main nokw; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/wgrad_$v.so"; fi
  echo -n "bench [$v]: "; env $L $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
done
