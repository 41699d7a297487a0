#!/bin/bash
# round 5, call 5: what reference train.py's own defaults cost (optimizer, DistributedDataParallel) on one box; MAS_SAVE_ACT at batch 32 and at the 192 per GPU
# the reference hints at; joules per launch of the Upsample convolution in both forms; Adam edge-case test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_adam.py -q 2>&1 | tail -2
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step | optimizer:', d['config']['optimizer'], '| parallelism:', d['config']['parallelism'], '| peak GiB', d['peak_memory_gib'], '| batch', d['config']['per_gpu_batch'])"; }
{
echo "# same box, same build, bench.py --no-cpu-baseline --no-also --no-encoder-stack (20 timed steps after 15 warm-up steps), B = 32"
for opt in mas torch torch-default; do echo "== --optimizer $opt"; timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --optimizer $opt 2>/dev/null | line; done
for dp in mas ddp; do echo "== one rank with the N>1 code path forced (MAS_BENCH_FORCE_DDP=1: RCCL process group of world size 1, SyncBatchNorm, reducer), --dp $dp"
  MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --dp $dp 2>/dev/null | line; done
echo "== --dp ddp --optimizer torch-default: reference train.py:32,61 with no override at all"; MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --dp ddp --optimizer torch-default 2>/dev/null | line
} > $O/dropin_defaults.txt 2>&1; cat $O/dropin_defaults.txt
{
echo "# activation memory (VERDICT r4 next #9): allocator high-water mark and step time, same box"
for b in 32 192; do for sa in 1 0; do
  st="--steps 20 --warmup 15"; [ $b = 192 ] && st="--steps 4 --warmup 3"
  echo "== batch $b MAS_SAVE_ACT=$sa"; MAS_SAVE_ACT=$sa timeout 600 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --batch $b $st 2>/dev/null | line
done; done
echo "== batch 32 MAS_GN_MATERIALIZE=0 (fused loaders everywhere)"; MAS_GN_MATERIALIZE=0 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | line
} > $O/save_act.txt 2>&1; cat $O/save_act.txt
timeout 600 python tools/probes/energy_budget.py up2 5 > $O/energy_up2.txt 2>&1; grep -v "^      \|^HIP\|^sources" $O/energy_up2.txt | cut -c1-240
