#!/bin/bash
# round 6, call 41: stress the side-stream weight gradient for run-to-run differences
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_41; mkdir -p $O
MAS_WGRAD_STREAM=1 REPS=150 timeout 600 python tools/experiments/side_stream_stress.py 2>&1 | grep -v Warn | tee $O/stress1.txt | head -40
MAS_WGRAD_STREAM=0 REPS=150 timeout 600 python tools/experiments/side_stream_stress.py 2>&1 | grep -v Warn | tee $O/stress0.txt | head -40
