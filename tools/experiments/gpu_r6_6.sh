#!/bin/bash
# round 6, call 6: the whole GPU suite + smoke on the mid-round tree (BN fp64 / eval autograd / fused LeakyReLU, own loss_seg, XCD bands, colsum hand-off)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_6; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=15 > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
