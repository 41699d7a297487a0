#!/bin/bash
# round 6, call 8: where the causal attention BACKWARD's time goes -- ablation builds of attn_bwd_dkv_bf16_kernel / attn_bwd_dq_bf16_kernel
# (B=8, H=16, S=1536, hd=64; -DFB_ABL_*), per-kernel durations from rocprofv3 --kernel-trace --stats around tools/kbench.py attn
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_8; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
cd /tmp
for v in "" fb_noexp fb_nosoftmax fb_nosdp fb_noout fb_nodma fb_empty fb_sdponly fb_softmaxonly fb_outonly fb_dmaonly fb_dkv1 fb_dkv3 fb_dq2 fb_dq4 ""; do
  echo "== variant ${v:-shipped}"
  if [ -n "$v" ]; then export MAS_HIP_LIB=$V/$v.so; else unset MAS_HIP_LIB; fi
  rm -rf /tmp/ab_$v
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ab_x -o ab -- python $GRAFT_REPO_ROOT/tools/kbench.py attn --n 8 --iters 50 > /tmp/ab.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/ab_x -name "*.db" | head -1) /tmp/ab_sum.txt > /dev/null 2>&1
  grep -E "attn_bwd_dkv_bf16|attn_bwd_dq_bf16|attn_bwd_delta|attn_causal_fwd" /tmp/ab_sum.txt | awk '{printf "   %-8s calls %5s avg_us %8s   %s\n", $1, $2, $4, $7}'
  rm -rf /tmp/ab_x
done | tee $GRAFT_REPO_ROOT/$O/attn_bwd_ablation.txt
