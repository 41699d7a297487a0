#!/bin/bash
# round 6, call 32: why does the N>1 code path (world size 1, RCCL) lose with the side-stream weight gradient?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_32; mkdir -p $O
for cfg in "0 0 mas" "1 -1 mas" "1 0 mas" "1 224 mas" "0 0 ddp" "1 -1 ddp"; do set -- $cfg
  MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_STREAM=$1 MAS_WGRAD_CUS=$2 timeout 300 python bench.py --dp $3 --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('FORCE_DDP dp=$3 STREAM=$1 CUS=$2', d['ms_per_step'], d['final_loss'])"
done | tee $O/ddp_ab.txt
cd /tmp
MAS_BENCH_FORCE_DDP=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/dd -o dd -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/dd.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/dd -name "*.db" | head -1) $O/trace_ddp_stream1.txt > /dev/null; head -14 $O/trace_ddp_stream1.txt | cut -c1-150; grep -i "nccl\|rccl\|CatArray\|AllReduce" $O/trace_ddp_stream1.txt | head -5 | cut -c1-150
