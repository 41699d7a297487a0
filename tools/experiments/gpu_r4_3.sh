#!/bin/bash
# round 4, session 3: the round-4 parity tests (decoder backward under the reference's drec, round-3 kernels at their benched shapes, fused statistics with tiles > grid)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_wide.py tests/test_gpu_parity_r3.py -x -q -s > $O/pytest_parity_r4.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_parity_r4.txt
grep -v "^  warn\|Warning\|amdgpu.ids" $O/pytest_parity_r4.txt | tail -90 | cut -c1-230
