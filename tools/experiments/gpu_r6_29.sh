#!/bin/bash
# round 6, call 29: side-stream weight gradient with a grid sized for fewer CUs (free CUs for the GroupNorm launches beside it)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_29; mkdir -p $O
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 240" "1 224" "1 192" "0 224"; do set -- $cfg
  MAS_WGRAD_STREAM=$1 MAS_WGRAD_CUS=$2 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('STREAM=$1 CUS=$2', d['ms_per_step'], d['final_loss'])"
done; done | tee $O/step_ab.txt
cd /tmp
MAS_WGRAD_STREAM=1 MAS_WGRAD_CUS=224 timeout 400 rocprofv3 --kernel-trace -d /tmp/ov -o ov -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/ov.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/overlap_timeline.py $(find /tmp/ov -name "*.db" | head -1) > $O/timeline224.txt 2>&1; head -14 $O/timeline224.txt
