#!/bin/bash
# round-3 GPU call 34: wide conv residual epilogue: both rows' residual requested up front (main) vs row by row (-DW_RES_SEQ)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity_r3.py -m gpu -x -q 2>&1 | tail -2
KB="timeout 120 python tools/kbench.py"
for v in main wide_resseq main wide_resseq; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
  echo "== [$v]"
  env $L $KB conv_fwd --n 32 --c 128 --hw 256 --res 1 2>&1 | tail -1
  env $L $KB conv_fwd --n 32 --c 128 --hw 128 --res 1 2>&1 | tail -1
  env $L $KB conv_fwd --n 32 --c 256 --hw 64 --res 1 2>&1 | tail -1
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in main wide_resseq main wide_resseq; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
  echo -n "bench [$v]: "; env $L $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
