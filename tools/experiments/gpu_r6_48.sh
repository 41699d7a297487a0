#!/bin/bash
# round 6, call 48: run-to-run stress in other geometries / modes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_48; mkdir -p $O
for mode in b8 r512 fp32; do MODE=$mode REPS=300 timeout 900 python tools/experiments/variant_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return " ; done | tee $O/variants.txt
MAS_GN_MATERIALIZE=0 MODE=fused REPS=300 timeout 900 python tools/experiments/variant_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return " | tee -a $O/variants.txt
