#!/bin/bash
# round 6, call 31: the new default (MAS_WGRAD_STREAM=1, MAS_WGRAD_CUS=-1): full suite, smoke, determinism, step A/B against the one-stream form, N>1 path at world size 1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_31; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python tools/probes/determinism.py 2>&1 | tail -3 | tee $O/determinism.txt
for rep in 1 2 3; do for b in 0 1; do
  MAS_WGRAD_STREAM=$b timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_WGRAD_STREAM=$b', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['final_loss'])"
done; done | tee $O/step_ab.txt
for b in 0 1; do MAS_WGRAD_STREAM=$b MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('FORCE_DDP MAS_WGRAD_STREAM=$b', d['ms_per_step'], d['final_loss'])"; done | tee -a $O/step_ab.txt
