#!/bin/bash
# round 6, call 46: run-to-run stress of the transformer step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_46; mkdir -p $O
TRIALS=60 STEPS=6 timeout 1500 python tools/experiments/transformer_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/tr.txt | tail -15
