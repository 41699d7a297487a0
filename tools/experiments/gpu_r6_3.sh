#!/bin/bash
# round 6, call 3: re-run of the three tests call 2 failed + the HBM-side traffic of whole training steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_3; mkdir -p $O
timeout 900 python -m pytest -m gpu -q -rP --timeout 600 tests/test_gpu_bn.py tests/test_gpu_losses.py > $O/pytest.txt 2>&1
grep -E "passed|failed|rel-L2|disc logits" $O/pytest.txt | tail -20
bash tools/step_traffic.sh $O/traffic 3 2 2>&1 | tail -45
cat $O/traffic/counters_available.txt | tr '\n' ' '
