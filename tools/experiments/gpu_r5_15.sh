#!/bin/bash
# round 5, call 15: causal attention forward, finer ablations (what does an EMPTY key loop cost, and each part alone)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_15; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
for v in "" fa_empty fa_qkonly fa_softmaxonly fa_pvonly fa_dmaonly fa_noqk ""; do
  echo "== variant ${v:-shipped}"
  if [ -n "$v" ]; then export MAS_HIP_LIB=$V/$v.so; else unset MAS_HIP_LIB; fi
  timeout 120 python tools/kbench.py attn --n 8 --iters 200 2>&1 | grep "^attn fwd B"
done | tee $O/attn_ablation2.txt
