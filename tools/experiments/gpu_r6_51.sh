#!/bin/bash
# round 6, call 51: final soak of the benched step on the final tree (two streams, the default; then one stream)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_51; mkdir -p $O
MAS_WGRAD_STREAM=1 CHECK=0 TRIALS=400 timeout 2400 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/s1.txt
MAS_WGRAD_STREAM=0 CHECK=0 TRIALS=200 timeout 1500 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/s0.txt
