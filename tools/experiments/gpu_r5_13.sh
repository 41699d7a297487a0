#!/bin/bash
# round 5, call 13: causal attention forward without the per-tile DMA wait (shipped build) and with fragment reads ahead of the MFMAs (FA_PIPE) at 4 / 3 / 2 work-groups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_13; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
for v in "" fa_pipe4 fa_pipe3 fa_fix3 fa_pipe2 ""; do
  echo "== variant ${v:-shipped}"
  if [ -n "$v" ]; then export MAS_HIP_LIB=$V/$v.so; else unset MAS_HIP_LIB; fi
  timeout 120 python tools/kbench.py attn --n 8 --iters 200 2>&1 | grep "^attn fwd"
  timeout 200 python -m pytest tests/test_gpu_transformer.py -x -q -k "attention" 2>&1 | tail -1
done | tee $O/attn_pipe.txt
