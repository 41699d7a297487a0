#!/bin/bash
# round 4, session 25: Upsample + conv in its sub-pixel form (conv3x3_stream NT = 4): parity, kbench-like timing, model tests, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_25; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_up2.py -x -q -rP > $O/pytest_up2.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_up2.txt; grep "forward max\|passed\|failed\|rc=\|Error\|^E " $O/pytest_up2.txt | head -20 | cut -c1-220
{ for u in 1 0; do for sh in "128 128" "256 64" "512 32" "512 16"; do set -- $sh
  MAS_CONV_UP2=$u timeout 120 python tools/kbench.py conv_fwd --n 32 --c $1 --hw $2 --ups 1 --iters 30 2>&1 | grep "^conv_fwd" | sed "s/^/up2=$u /"
done; done; } > $O/kbench.txt 2>&1; cat $O/kbench.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r4.py tests/test_gpu_parity_r3.py -x -q > $O/pytest_model.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_model.txt; tail -3 $O/pytest_model.txt | cut -c1-200
for u in 1 0 1 0; do
MAS_CONV_UP2=$u timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('up2=$u', d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('avg_launch_ms'), 'clock', r.get('sustained_clock_mhz'), 'MHz')"
done > $O/bench.txt 2>&1; cat $O/bench.txt
