#!/bin/bash
# round 5, call 7: the weight gradient of Upsample + conv in the phase form (conv_wgrad_dma KS = 2 + mas_wgrad_reduce_up2): parity, kbench, step A/B;
# then the whole GPU suite (N>1 defaults changed; bf16 yardstick at 1.2x)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_up2.py -q -s > $O/pytest_up2.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_up2.txt
grep -v "^  warn\|Warning\|amdgpu.ids" $O/pytest_up2.txt | grep "wgrad\|passed\|failed\|FAIL\|Error\|assert\|rc=" | tail -40 | cut -c1-250
for v in 1 0; do
  for cfg in "128 128" "256 64" "512 32"; do set -- $cfg
    MAS_CONV_UP2_WGRAD=$v timeout 120 python tools/kbench.py wgrad --c $1 --hw $2 --ups 1 --iters 200 2>&1 | tail -1 | sed "s/^/MAS_CONV_UP2_WGRAD=$v /"
  done
done > $O/kbench_up2_wgrad.txt 2>&1; cat $O/kbench_up2_wgrad.txt
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step')"; }
for rep in 1 2; do for v in 1 0; do echo "== MAS_CONV_UP2_WGRAD=$v (rep $rep)"; MAS_CONV_UP2_WGRAD=$v timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | line; done; done > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-250
