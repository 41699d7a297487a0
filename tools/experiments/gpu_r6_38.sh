#!/bin/bash
# round 6, call 38: side stream with the overlap probe + GPU_MAX_HW_QUEUES default 8; explicit 4 must be refused under a process group
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_38; mkdir -p $O
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python tools/experiments/reducer_ab.py --no-cpu-baseline --no-also --no-encoder-stack 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$label', d['ms_per_step'])"
  grep -o "one spin kernel[^)]*" $O/err.txt | sed "s/^/   refused: /"
}
for q in 4 default; do
  if [ $q = default ]; then Q="MAS_NOP=1"; else Q="GPU_MAX_HW_QUEUES=$q"; fi
  for st in 0 1; do
    run "queues=$q plain STREAM=$st" $Q MODE=full MAS_WGRAD_STREAM=$st
    run "queues=$q pgonly STREAM=$st" $Q MODE=pgonly MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_STREAM=$st
    run "queues=$q reducer STREAM=$st" $Q MODE=full MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_STREAM=$st
  done
done | tee $O/ab.txt
run "queues=default reducer STREAM=1 min_elems=2^23" MODE=full MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_STREAM=1 MAS_WGRAD_STREAM_MIN_ELEMS=8388608 | tee -a $O/ab.txt
for st in 0 1; do
  MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_STREAM=$st timeout 300 python bench.py --dp ddp --no-cpu-baseline --no-also --no-encoder-stack 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('queues=default ddp STREAM=$st', d['ms_per_step'])"
  grep -o "one spin kernel[^)]*" $O/err.txt | sed "s/^/   refused: /"
done | tee -a $O/ab.txt
