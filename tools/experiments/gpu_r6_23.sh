#!/bin/bash
# round 6, call 23: run-to-run determinism of the final tree (XCD-banded walks, fp64 BatchNorm sums, LayerNorm pair, GELU colsum)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_23; mkdir -p $O
timeout 600 python tools/probes/determinism.py > $O/determinism.txt 2>&1; tail -12 $O/determinism.txt
