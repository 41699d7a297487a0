#!/bin/bash
# round 5, call 6: where the N>1 code path loses 3-4 ms at world size 1: MAS_WGRAD_OVERSUB (2 work-groups per CU in the weight gradients), the reducer's
# accumulate-in-place scheme against the flatten copy, DistributedDataParallel; kernel trace of the forced-DDP step; bf16 yardstick tests with their printed ratios
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_6; mkdir -p $O
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step | peak GiB', d['peak_memory_gib'])"; }
{
echo "== plain (no process group)"; timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | line
for os_ in 2 1; do for dp in mas ddp; do for stn in 0 1; do
  [ $dp = ddp ] && [ $stn = 1 ] && continue
  echo "== FORCE_DDP --dp $dp MAS_WGRAD_OVERSUB=$os_ MAS_BENCH_REDUCER_SET_TO_NONE=$stn"
  MAS_BENCH_FORCE_DDP=1 MAS_WGRAD_OVERSUB=$os_ MAS_BENCH_REDUCER_SET_TO_NONE=$stn timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --dp $dp 2>/dev/null | line
done; done; done
echo "== plain, MAS_WGRAD_OVERSUB=2 alone"; MAS_WGRAD_OVERSUB=2 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | line
} > $O/ddp_breakdown.txt 2>&1; cat $O/ddp_breakdown.txt
cd /tmp && MAS_BENCH_FORCE_DDP=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_ddp -o vq -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/pf_ddp.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $(find /tmp/pf_ddp -name "*.db" | head -1) $O/kernel_trace_forced_ddp.txt > /dev/null 2>&1; head -40 $O/kernel_trace_forced_ddp.txt | cut -c1-170
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -q -s -k "backward" > $O/pytest_yardstick.txt 2>&1; grep "rel-L2\|passed\|failed" $O/pytest_yardstick.txt | cut -c1-200 | tail -80
