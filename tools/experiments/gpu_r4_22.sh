#!/bin/bash
# round 4, session 22: run-to-run reproducibility of the final loss with either optimizer (same seeds)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_22; mkdir -p $O
for o in mas mas mas torch torch torch mas; do
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also --optimizer $o 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('optimizer=$o', d['ms_per_step'], 'ms/step; loss', d['final_loss'], 'spread', d['replica_weight_checksum_spread'])"
done > $O/loss.txt 2>&1; cat $O/loss.txt
