#!/bin/bash
# round 6, call 37: is GPU_MAX_HW_QUEUES still honoured when set after `import torch` (before the first HIP call)?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_37; mkdir -p $O
for late in import avail; do
  for st in 0 1; do
    LATE_QUEUES=$late MAS_BENCH_FORCE_DDP=1 MODE=full MAS_WGRAD_STREAM=$st timeout 300 python tools/experiments/reducer_ab.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('late=$late full STREAM=$st', d['ms_per_step'])"
  done
done | tee $O/ab.txt
