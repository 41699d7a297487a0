#!/bin/bash
# round 5, call 11: GroupNorm backward variants (raw second sum, occupancy bounds) on the probe shapes, block budget 8192
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_11; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
for v in "" gn_raw gn_raw_w5 gn_raw_w6 gn_w6 gn_raw_w8; do
  echo "== variant ${v:-shipped}"
  if [ -n "$v" ]; then export MAS_HIP_LIB=$V/$v.so; else unset MAS_HIP_LIB; fi
  MAS_GN_ACT_BLOCKS=8192 MAS_GN_APPLY_BLOCKS=8192 timeout 300 python tools/probes/gn_blocks_probe.py child 2>&1 | grep "^n="
done | tee $O/gn_variants.txt
