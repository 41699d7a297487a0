#!/bin/bash
# round 6, call 13: GELU backward with the lin1 bias gradient: tests + transformer step A/B (MAS_GELU_COLSUM=0 / 1) + kernel rows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_13; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --timeout 600 tests/test_gpu_transformer.py tests/test_gpu_sampling.py > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2 3; do for b in 0 1; do
  MAS_GELU_COLSUM=$b timeout 300 python bench.py --workload transformer 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_GELU_COLSUM=$b', d['ms_per_step'], d['value'])"
done; done | tee $O/step_ab.txt
cd /tmp
for b in 0 1; do
MAS_GELU_COLSUM=$b timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_g$b -o tr -- python $GRAFT_REPO_ROOT/bench.py --workload transformer --steps 4 --warmup 1 > /tmp/pf_g$b.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pf_g$b -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/trace_gelu$b.txt > /dev/null
grep -i "gelu\|colsum\|fold_rows" $GRAFT_REPO_ROOT/$O/trace_gelu$b.txt | cut -c1-140
done
