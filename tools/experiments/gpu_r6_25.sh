#!/bin/bash
# round 6, call 25: weight gradients on a side stream beside the GroupNorm backward passes (MAS_WGRAD_STREAM=1): parity + step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_25; mkdir -p $O
MAS_WGRAD_STREAM=1 timeout 900 python -m pytest -m gpu -q --timeout 600 tests/test_gpu_parity_r5.py tests/test_gpu_parity_r3.py tests/test_gpu_model.py > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2 3; do for b in 0 1; do
  MAS_WGRAD_STREAM=$b timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_WGRAD_STREAM=$b', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['final_loss'])"
done; done | tee $O/step_ab.txt
