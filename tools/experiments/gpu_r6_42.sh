#!/bin/bash
# round 6, call 42: bench-like loop, repeated from the same state: does the end state ever differ, and which gradient first?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_42; mkdir -p $O
MAS_PACK_SIDE=0 MAS_WGRAD_STREAM=1 CHECK=0 TRIALS=40 timeout 900 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/s1_c0.txt
MAS_PACK_SIDE=0 MAS_WGRAD_STREAM=1 CHECK=1 TRIALS=40 timeout 900 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/s1_c1.txt
MAS_PACK_SIDE=0 MAS_WGRAD_STREAM=0 CHECK=0 TRIALS=40 timeout 900 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/s0_c0.txt
