#!/bin/bash
# round 6, call 11: is the backward's tile-DMA LATENCY exposed (a deeper ring would help) or only its issue cost?  -DFB_ABL_NOWAIT drops the vmcnt wait (timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_11; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
cd /tmp
for v in "" fb_nowait fb_nodma ""; do
  echo "== variant ${v:-shipped}"
  if [ -n "$v" ]; then export MAS_HIP_LIB=$V/$v.so; else unset MAS_HIP_LIB; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ab_x -o ab -- python $GRAFT_REPO_ROOT/tools/kbench.py attn --n 8 --iters 50 > /tmp/ab.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/ab_x -name "*.db" | head -1) /tmp/ab_sum.txt > /dev/null 2>&1
  grep -E "attn_bwd_dkv_bf16|attn_bwd_dq_bf16" /tmp/ab_sum.txt | awk '{printf "   calls %5s avg_us %8s   %s\n", $2, $4, $7}'
  rm -rf /tmp/ab_x
done | tee $GRAFT_REPO_ROOT/$O/nowait.txt
