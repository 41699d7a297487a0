#!/bin/bash
# round 5, call 12: where the causal attention forward's time goes -- ablation builds of attn_causal_fwd_bf16_v2_kernel (B=8, H=16, S=1536, hd=64)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_12; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
for v in "" fa_noexp fa_nosoftmax fa_nopv fa_nosoftmax_nopv fa_nobar fa_nodma_nobar ""; do
  echo "== variant ${v:-shipped}"
  if [ -n "$v" ]; then export MAS_HIP_LIB=$V/$v.so; else unset MAS_HIP_LIB; fi
  timeout 120 python tools/kbench.py attn --n 8 --iters 200 2>&1 | grep "^attn fwd B"
done | tee $O/attn_ablation.txt
