#!/bin/bash
# round 6, call 35: which part of the reducer path loses the side-stream gain (tools/experiments/reducer_ab.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_35; mkdir -p $O
for mode in full pgonly nohook noar nocat; do
  for st in 0 1; do
    MODE=$mode MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_STREAM=$st timeout 300 python tools/experiments/reducer_ab.py --no-cpu-baseline --no-also --no-encoder-stack 2>$O/err_${mode}_$st.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$mode STREAM=$st', d['ms_per_step'])"
  done
done | tee $O/ab.txt
tail -3 $O/err_noar_1.txt
