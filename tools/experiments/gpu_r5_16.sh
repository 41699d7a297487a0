#!/bin/bash
# round 5, call 16: attention forward v3 (score MFMAs of the next tile between the softmax instructions) against v2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_16; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
for cfg in "v2:MAS_ATTN_FWD_V3=0:" "v3_3wg:MAS_ATTN_FWD_V3=1:" "v3_2wg:MAS_ATTN_FWD_V3=1:$V/fa_v3w2.so" "v2:MAS_ATTN_FWD_V3=0:"; do
  IFS=: read name envv libv <<< "$cfg"
  echo "== $name"
  if [ -n "$libv" ]; then export MAS_HIP_LIB=$libv; else unset MAS_HIP_LIB; fi
  env $envv timeout 120 python tools/kbench.py attn --n 8 --iters 200 2>&1 | grep "^attn fwd"
  env $envv timeout 200 python -m pytest tests/test_gpu_transformer.py -x -q -k "attention" 2>&1 | tail -1
done | tee $O/attn_v3.txt
