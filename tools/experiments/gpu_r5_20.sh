#!/bin/bash
# round 5, call 20: attention forward with the cheaper DMA issue (M0 bases in SGPRs, no bounds select on full tiles), 3 work-groups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_20; mkdir -p $O
for rep in 1 2 3; do timeout 120 python tools/kbench.py attn --n 8 --iters 200 2>&1 | grep "^attn fwd"; done | tee $O/attn_issue.txt
timeout 300 python -m pytest tests/test_gpu_transformer.py -x -q -k "attention" 2>&1 | tail -1 | tee -a $O/attn_issue.txt
