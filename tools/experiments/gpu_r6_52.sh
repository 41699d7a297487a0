#!/bin/bash
# round 6, call 52: weight-gradient grid size per layer size (experiment knob MAS_WGRAD_CUS_BY_SIZE=elems:cus_small:cus_large)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_52; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', d['ms_per_step'], d.get('final_loss'))"; }
for rep in 1 2; do
  run MAS_NOP=1
  run MAS_WGRAD_CUS_BY_SIZE=8388608:0:-1
  run MAS_WGRAD_CUS_BY_SIZE=8388608:224:-1
  run MAS_WGRAD_CUS_BY_SIZE=33554432:0:-1
  run MAS_WGRAD_CUS_BY_SIZE=33554432:224:-1
  run MAS_WGRAD_CUS_BY_SIZE=67108864:-1:176
  run MAS_WGRAD_CUS_BY_SIZE=67108864:-1:208
done | tee $O/ab.txt
