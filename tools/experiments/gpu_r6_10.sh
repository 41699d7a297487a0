#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_10; mkdir -p $O
MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack > $O/force_ddp.json 2> $O/force_ddp.err; tail -5 $O/force_ddp.err; cut -c1-400 $O/force_ddp.json
MAS_BENCH_FORCE_DDP=1 MAS_SYNCBN_EXCHANGE_AT_WORLD_1=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack > $O/force_ddp_x.json 2> $O/force_ddp_x.err; tail -5 $O/force_ddp_x.err; cut -c1-400 $O/force_ddp_x.json
timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | cut -c1-200
