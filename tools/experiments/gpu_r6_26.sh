#!/bin/bash
# round 6, call 26: timeline of the side-stream weight gradient (do wgrad and gn_bwd really run side by side?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_26; mkdir -p $O
cd /tmp
MAS_WGRAD_STREAM=1 timeout 400 rocprofv3 --kernel-trace -d /tmp/ov -o ov -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/ov.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/overlap_timeline.py $(find /tmp/ov -name "*.db" | head -1) > $O/timeline.txt 2>&1; head -40 $O/timeline.txt
