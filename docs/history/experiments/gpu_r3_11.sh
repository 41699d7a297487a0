#!/bin/bash
# round-3 GPU call 14: attention forward v2 (DMA staging, swizzled LDS, 4 work-groups per CU): parity + kbench + transformer step A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_11; mkdir -p $O
cd $R
echo "== pytest transformer"
timeout 900 python -m pytest tests/test_gpu_transformer.py tests/test_gpu_sampling.py tests/test_gpu_parity_r3.py -m gpu -q -k "attention or make_a_scene or sampling or cached" 2>&1 | tail -4
KB="timeout 120 python tools/kbench.py"
for v in 1 0; do
  echo "== kbench attn MAS_ATTN_FWD_V2=$v"
  for n in 8 16 32; do MAS_ATTN_FWD_V2=$v $KB attn --n $n 2>&1 | grep -v amdgpu.ids | head -1; done
done
for v in 1 0 1 0; do
  echo "== bench transformer MAS_ATTN_FWD_V2=$v"
  MAS_ATTN_FWD_V2=$v timeout 300 python bench.py --workload transformer --steps 12 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.1f tok/s  %.3f ms/step  attn fwd %.4f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
