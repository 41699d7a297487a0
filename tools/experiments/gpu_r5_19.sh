#!/bin/bash
# round 5, call 19: attention forward with complementary query blocks paired per work-group (MAS_ATTN_PAIR=1) against one block per work-group
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_19; mkdir -p $O
V=$GRAFT_REPO_ROOT/make-a-scene_amd/csrc/build/variants
for pair in 0 1 0 1; do
  echo "== MAS_ATTN_PAIR=$pair"
  MAS_ATTN_PAIR=$pair timeout 120 python tools/kbench.py attn --n 8 --iters 200 2>&1 | grep "^attn fwd"
done | tee $O/attn_pair.txt
MAS_ATTN_PAIR=1 timeout 300 python -m pytest tests/test_gpu_transformer.py -x -q -k "attention" 2>&1 | tail -1 | tee -a $O/attn_pair.txt
for pair in 0 1; do echo "== trace build, MAS_ATTN_PAIR=$pair"; MAS_ATTN_PAIR=$pair MAS_HIP_LIB=$V/fa_trace.so timeout 120 python tools/kbench.py attn --n 8 --iters 50 2>&1 | grep "^attn fwd\|^  that"; done | tee -a $O/attn_pair.txt
