#!/bin/bash
# round 6, call 62: reservoir permutations: 2 = torch.randperm(n, device=cuda) (rounds 1-6), 0 = CPU permutation + blocking copy (the reference's), 1 = CPU permutation + pinned non-blocking copy
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_62; mkdir -p $O
for rep in 1 2 3 4; do
  for on in 2 1 0; do
    MAS_RESERVOIR_ASYNC=$on timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_RESERVOIR_ASYNC=$on', d['ms_per_step'], d['value'], d.get('final_loss'))"
  done
done | tee $O/ab.txt
