#!/bin/bash
# round-3 GPU call 20: conv kernels without SLP packing of fp32 VALU (v_pk_*_f32 beside MFMAs): kbench + step A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
for v in main wgrad_noslp conv_noslp main conv_noslp; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
  echo "== [$v]"
  env $L $KB wgrad --n 32 --c 128 --hw 256 2>&1 | tail -1
  env $L $KB conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1
  env $L $KB conv_fwd --n 32 --c 128 --hw 256 --res 1 2>&1 | tail -1
  env $L $KB conv_fwd --n 32 --c 512 --hw 32 2>&1 | tail -1
  env $L $KB conv_fwd --n 32 --c 512 --hw 16 2>&1 | tail -1
done
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in main conv_noslp main conv_noslp; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/$v.so"; fi
  echo -n "bench [$v]: "; env $L $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
