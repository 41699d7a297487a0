#!/bin/bash
# round 6, call 9: SyncBatchNorm exchange over RCCL (world size 1, forced), N>1 dry runs with the new default, LayerNorm pair (shipped form) A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_9; mkdir -p $O
timeout 1200 python -m pytest -m gpu -q -rs --timeout 600 tests/test_gpu_bn.py tests/test_gpu_dp.py tests/test_gpu_transformer.py > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for rep in 1 2 3; do for b in 0 1; do
  MAS_LN_PAIR=$b timeout 300 python bench.py --workload transformer 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_LN_PAIR=$b', d['ms_per_step'], d['value'])"
done; done | tee $O/step_ab.txt
MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('FORCE_DDP (nccl, world 1)', d['ms_per_step'], d['config']['parallelism'])" | tee -a $O/step_ab.txt
