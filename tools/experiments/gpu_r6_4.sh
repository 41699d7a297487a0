#!/bin/bash
# round 6, call 4: step traffic again with the request-size counters (TCC_EA0_RDREQ_64B / _128B exist on this rocprofv3) + BN tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_4; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --timeout 600 tests/test_gpu_bn.py > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
bash tools/step_traffic.sh $O/traffic 3 2 2>&1 | tail -60
python3 tools/step_traffic.py $O/traffic 3 2 --json $O/step_traffic.json > /dev/null
