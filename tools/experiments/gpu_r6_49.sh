#!/bin/bash
# round 6, call 49: the general weight-gradient kernels in slab mode (no fp32 atomics left): full suite, then the run-to-run stress of the paths that had them
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_49; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -8 > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
REPS=200 timeout 900 python tools/experiments/misc_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return " | tee $O/misc.txt
MODE=fp32 REPS=200 timeout 900 python tools/experiments/variant_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return " | tee $O/fp32.txt
MAS_GN_MATERIALIZE=0 MODE=fused REPS=200 timeout 900 python tools/experiments/variant_stress.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return " | tee $O/fused.txt
timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['ms_per_step'], d['value'], d.get('final_loss'))" | tee $O/bench.txt
