#!/bin/bash
# round 6, call 34: side-stream weight gradient only for large layers (MAS_WGRAD_STREAM_MIN_ELEMS): plain / GradReducer / DDP at world size 1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_34; mkdir -p $O
for thr in 0 8388608 33554432 67108864; do
  for mode in plain mas ddp; do
    if [ $mode = plain ]; then E=""; D="mas"; else E="MAS_BENCH_FORCE_DDP=1"; D=$mode; fi
    env $E MAS_WGRAD_STREAM=1 MAS_WGRAD_STREAM_MIN_ELEMS=$thr timeout 300 python bench.py --dp $D --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('STREAM=1 min_elems=$thr $mode', d['ms_per_step'])"
  done
done | tee $O/thr_ab.txt
for mode in plain mas ddp; do
    if [ $mode = plain ]; then E=""; D="mas"; else E="MAS_BENCH_FORCE_DDP=1"; D=$mode; fi
    env $E MAS_WGRAD_STREAM=0 timeout 300 python bench.py --dp $D --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('STREAM=0 $mode', d['ms_per_step'])"
done | tee -a $O/thr_ab.txt
