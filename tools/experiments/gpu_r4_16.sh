#!/bin/bash
# round 4, session 16: 8-row tiles of the stream kernel on small maps: parity (bitwise vs 16-row tiles), kbench at 512 @16^2, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream_th8.py -x -q -rP > $O/pytest_th8.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_th8.txt; grep "max err\|dgrad\|passed\|failed\|rc=\|Error\|assert" $O/pytest_th8.txt | head -20 | cut -c1-200
{ for t in 1 0; do for s in "512 16" "256 16" "512 8"; do set -- $s
  MAS_CONV_STREAM_TH8=$t timeout 120 python tools/kbench.py conv_fwd --n 32 --c $1 --hw $2 --iters 50 2>&1 | grep "^conv_fwd" | sed "s/^/th8=$t /"
  MAS_CONV_STREAM_TH8=$t timeout 120 python tools/kbench.py dgrad --n 32 --c $1 --hw $2 --iters 50 2>&1 | grep "^dgrad" | sed "s/^/th8=$t /"
done; done; } > $O/kbench.txt 2>&1; cat $O/kbench.txt
for t in 1 0 1 0; do
MAS_CONV_STREAM_TH8=$t timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('th8=$t', d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('avg_launch_ms'), 'clock', r.get('sustained_clock_mhz'), 'MHz; enc', d['encoder_stack']['fwd_ms'])"
done > $O/bench.txt 2>&1; cat $O/bench.txt
