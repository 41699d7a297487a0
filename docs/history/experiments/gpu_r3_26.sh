#!/bin/bash
# round-3 GPU call 31: GroupNorm backward walking the batch in Infinity-Cache-sized image groups (MAS_GN_BWD_CHUNK_MB)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for mb in 0 96 160 224 300 400; do
  echo "== MAS_GN_BWD_CHUNK_MB=$mb"
  MAS_GN_BWD_CHUNK_MB=$mb timeout 120 python tools/probes/gn_chunk_probe.py full 2>&1 | grep "^c="
done
MAS_GN_BWD_CHUNK_MB=160 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for mb in 0 160 224 0 160 224; do
  echo -n "bench [chunk_mb=$mb]: "; MAS_GN_BWD_CHUNK_MB=$mb $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
