#!/bin/bash
# round 4, session 23: ResnetBlock with its 1x1 shortcut as one autograd node (the skip gradient enters norm1's backward as dres)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_23; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py tests/test_gpu_dp.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt | cut -c1-200
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('avg_launch_ms'), 'clock', r.get('sustained_clock_mhz'), 'MHz; enc', d['encoder_stack']['fwd_ms'], 'mem', d.get('peak_memory_gib'))"
done > $O/bench.txt 2>&1; cat $O/bench.txt
