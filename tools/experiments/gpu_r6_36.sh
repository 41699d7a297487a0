#!/bin/bash
# round 6, call 36: HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) vs the side stream once RCCL has made its own streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_36; mkdir -p $O
for q in default 8 16; do
  for mode in plain pgonly full; do
    for st in 0 1; do
      if [ $q = default ]; then Q=""; else Q="GPU_MAX_HW_QUEUES=$q"; fi
      if [ $mode = plain ]; then F=""; M=full; else F="MAS_BENCH_FORCE_DDP=1"; M=$mode; fi
      env $Q $F MODE=$M MAS_WGRAD_STREAM=$st timeout 300 python tools/experiments/reducer_ab.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('queues=$q $mode STREAM=$st', d['ms_per_step'])"
    done
  done
done | tee $O/ab.txt
