#!/bin/bash
# round 6, call 47: run-to-run stress of the other paths + a long transformer run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_47; mkdir -p $O
REPS=400 timeout 1500 python tools/experiments/misc_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return " | tee $O/misc.txt | tail
TRIALS=300 STEPS=6 timeout 1500 python tools/experiments/transformer_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/tr.txt | tail -8
