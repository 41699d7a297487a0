#!/bin/bash
# round-3 GPU call 15: attention backward with DMA staging (dq: 3 waves per SIMD; dkv: 2 or 3): parity + kbench + transformer A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_12; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
echo "== pytest transformer (main build)"
timeout 900 python -m pytest tests/test_gpu_transformer.py tests/test_gpu_sampling.py tests/test_gpu_parity_r3.py -m gpu -q -k "attention or make_a_scene or sampling or cached" 2>&1 | tail -3
echo "== pytest attention (dkv3 variant)"
MAS_HIP_LIB=$V/attn_dkv3.so timeout 900 python -m pytest tests/test_gpu_transformer.py -m gpu -q -k "attention" 2>&1 | tail -2
KB="timeout 120 python tools/kbench.py"
for v in main olddkv dkv3; do
  echo "== kbench attn [$v]"
  if [ $v = main ]; then L=""; else L="MAS_HIP_LIB=$V/attn_$v.so"; fi
  for n in 8 16; do env $L $KB attn --n $n 2>&1 | grep -v amdgpu.ids | head -2; done
done
for v in main olddkv dkv3 main; do
  echo "== bench transformer [$v]"
  if [ $v = main ]; then L=""; else L="MAS_HIP_LIB=$V/attn_$v.so"; fi
  env $L timeout 300 python bench.py --workload transformer --steps 12 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.1f tok/s  %.3f ms/step  attn fwd %.4f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
