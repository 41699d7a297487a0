#!/bin/bash
# round 6, call 45: after the conv_fwd.hip stage-0 barrier fix: bench-like loop repeated from the same state, one stream and two
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_45; mkdir -p $O
MAS_PACK_SIDE=0 MAS_WGRAD_STREAM=0 CHECK=0 TRIALS=200 timeout 1500 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/s0.txt
MAS_PACK_SIDE=0 MAS_WGRAD_STREAM=1 CHECK=0 TRIALS=200 timeout 1500 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/s1.txt
MAS_PACK_SIDE=1 MAS_WGRAD_STREAM=1 CHECK=0 TRIALS=200 timeout 1500 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/s1p1.txt
