#!/bin/bash
# round 6, call 55: side-stream priority and the apply pass's grid budget under the two-stream schedule
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_55; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', d['ms_per_step'], d.get('final_loss'))"; }
for rep in 1 2; do
  run MAS_NOP=1
  run MAS_WGRAD_STREAM_PRIORITY=-1
  run MAS_GN_APPLY_BLOCKS=4096
  run MAS_GN_APPLY_BLOCKS=16384
  run MAS_GN_APPLY_BLOCKS=32768
  run MAS_WGRAD_STREAM_PRIORITY=-1 MAS_WGRAD_CUS=208
  run MAS_WGRAD_STREAM_PRIORITY=-1 MAS_WGRAD_CUS=176
done | tee $O/ab.txt
