#!/bin/bash
# round 6, call 7: LayerNorm pair -- parity tests, transformer step A/B (MAS_LN_PAIR=0 / 1), kernel trace of the paired step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_7; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --timeout 600 tests/test_gpu_transformer.py tests/test_gpu_sampling.py > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for rep in 1 2; do for b in 0 1; do
  MAS_LN_PAIR=$b timeout 300 python bench.py --workload transformer 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_LN_PAIR=$b', d['ms_per_step'], d['value'])"
done; done | tee $O/step_ab.txt
cd /tmp
for b in 0 1; do
MAS_LN_PAIR=$b timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_tr$b -o tr -- python $GRAFT_REPO_ROOT/bench.py --workload transformer --steps 4 --warmup 1 > /tmp/pf_tr$b.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pf_tr$b -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/kernel_trace_transformer_pair$b.txt > /dev/null
grep -i "layernorm\|fold_rows" $GRAFT_REPO_ROOT/$O/kernel_trace_transformer_pair$b.txt | cut -c1-150
done
