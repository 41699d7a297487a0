#!/bin/bash
# round 6, call 44: forward-only stress with per-module checksums
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_44; mkdir -p $O
GRAD=1 REPS=2500 timeout 900 python tools/experiments/fwd_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/fwd_g1.txt
GRAD=0 REPS=2500 timeout 900 python tools/experiments/fwd_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/fwd_g0.txt
