#!/bin/bash
# round 6, call 14: step traffic of the FINAL tree (XCD bands on, fp64 BN sums, ...) -> profiles/r06_step_traffic.{txt,json}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_14; mkdir -p $O
bash tools/step_traffic.sh $O/traffic 3 2 2>&1 | tail -70
python3 tools/step_traffic.py $O/traffic 3 2 --json $O/step_traffic.json > $O/step_traffic.txt
