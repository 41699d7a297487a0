#!/bin/bash
# round 6, call 40: is the final loss reproducible with MAS_PACK_SIDE=1?  (one of four runs in call 39 differed)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_40; mkdir -p $O
for rep in 1 2 3 4 5 6; do
  for ps in 1 0; do
    MAS_PACK_SIDE=$ps timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>$O/err_${ps}_$rep.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_PACK_SIDE=$ps', d['ms_per_step'], d['value'], d.get('final_loss'))"
    grep -o "one spin kernel[^)]*" $O/err_${ps}_$rep.txt | sed "s/^/   refused: /"
  done
done | tee $O/ab.txt
