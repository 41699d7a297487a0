#!/bin/bash
# round 6, call 30: sweep of MAS_WGRAD_CUS with the side-stream weight gradient
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_30; mkdir -p $O
for rep in 1 2; do
for cfg in "0 0" "1 208" "1 192" "1 176" "1 160" "1 128"; do set -- $cfg
  MAS_WGRAD_STREAM=$1 MAS_WGRAD_CUS=$2 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('STREAM=$1 CUS=$2', d['ms_per_step'], d['final_loss'], d['peak_memory_gib'])"
done; done | tee $O/step_ab.txt
