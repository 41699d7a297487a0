#!/bin/bash
# round 6, call 33: GradReducer bucket size with / without the side-stream weight gradient (N>1 code path at world size 1, RCCL)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_33; mkdir -p $O
for cfg in "0 128" "1 128" "0 25" "1 25" "0 512" "1 512"; do set -- $cfg
  MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_STREAM=$1 MAS_DP_BUCKET_MB=$2 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('GradReducer STREAM=$1 bucket=$2 MiB', d['ms_per_step'], d['final_loss'])"
done | tee $O/bucket_ab.txt
for b in 0 1; do MAS_WGRAD_STREAM=$b timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plain STREAM=$b', d['ms_per_step'])"; done | tee -a $O/bucket_ab.txt
