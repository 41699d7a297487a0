#!/bin/bash
# round 6, call 1: Winograd F(2x2,3x3) inner-loop probe (tools/probes/wino_loop.hip) + a baseline bench line of the round-5 tree on this pool
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_1; mkdir -p $O
[ -x tools/probes/wino_loop ] || bash tools/probes/build.sh > /dev/null 2>&1
{ timeout 120 tools/probes/wino_loop 64; timeout 120 tools/probes/wino_loop 64; } > $O/wino_loop.txt 2>&1
cat $O/wino_loop.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json
