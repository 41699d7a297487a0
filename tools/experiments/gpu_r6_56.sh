#!/bin/bash
# round 6, call 56: plain repetition of single launches of the general kernel, the fixed library and one built from the file before the fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_56; mkdir -p $O
for noise in 0 1; do
  echo "== fixed library, NOISE=$noise"; NOISE=$noise REPS=20000 timeout 600 python tools/experiments/conv_fwd_repeat.py 2>&1 | grep "^conv"
  echo "== library with conv_fwd.hip from before the fix, NOISE=$noise"; MAS_HIP_LIB=$GRAFT_REPO_ROOT/tmp_oldlib/libmas_hip_old.so NOISE=$noise REPS=20000 timeout 600 python tools/experiments/conv_fwd_repeat.py 2>&1 | grep "^conv"
done | tee $O/repeat.txt
