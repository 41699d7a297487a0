#!/bin/bash
# round 6, call 39: packed-weight refresh on the side stream beside the first convolutions of the forward (MAS_PACK_SIDE) + the new side-stream test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_39; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_model.txt; tail -3 $O/pytest_model.txt
for rep in 1 2 3; do
  for ps in 0 1; do
    MAS_PACK_SIDE=$ps timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_PACK_SIDE=$ps', d['ms_per_step'], d['value'], d['final_loss'] if 'final_loss' in d else '')"
  done
done | tee $O/ab.txt
MAS_PACK_SIDE=1 MAS_PACK_EARLY_BYTES=8388608 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_PACK_SIDE=1 early=8MB', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
MAS_PACK_SIDE=1 MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('reducer MAS_PACK_SIDE=1', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
MAS_PACK_SIDE=0 MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('reducer MAS_PACK_SIDE=0', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
