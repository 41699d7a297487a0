#!/bin/bash
# round 6, call 43: localise the rare run-to-run difference (forward output / gradient / parameter after Adam, per step), one stream
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_43; mkdir -p $O
MAS_PACK_SIDE=0 MAS_WGRAD_STREAM=0 CHECK=2 TRIALS=100 timeout 1500 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/s0_c2.txt
MAS_PACK_SIDE=0 MAS_WGRAD_STREAM=0 CHECK=1 TRIALS=100 timeout 1500 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/s0_c1.txt
