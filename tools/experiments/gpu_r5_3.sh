#!/bin/bash
# round 5, call 3: the whole GPU suite on the ABI v6 build; what reference train.py's own defaults cost (torch.optim.Adam as train.py:61 constructs it,
# DistributedDataParallel as train.py:32 wraps) against the bench defaults, same box; kernel trace of the step alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt | cut -c1-250
{
for opt in mas torch torch-default; do
  echo "== --optimizer $opt"; timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --optimizer $opt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['optimizer'], 'peak GiB', d['peak_memory_gib'])"
done
for dp in mas ddp; do
  echo "== one rank, the N>1 code path forced (MAS_BENCH_FORCE_DDP=1), --dp $dp"; MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --dp $dp 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['parallelism'])"
done
echo "== --dp ddp --optimizer torch-default (what train.py gets with no override)"; MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --dp ddp --optimizer torch-default 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
} > $O/dropin_defaults.txt 2>&1; cat $O/dropin_defaults.txt
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/pf_vq.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq_steps.txt > /dev/null 2>&1; head -30 $O/kernel_trace_vq_steps.txt | cut -c1-160
