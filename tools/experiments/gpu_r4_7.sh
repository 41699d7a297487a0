#!/bin/bash
# round 4, session 7: small-map GroupNorm kernels (parity, kbench, step A/B), full GPU suite after the knob pruning
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gn_coop.py -x -q > $O/pytest_gn.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gn.txt; tail -6 $O/pytest_gn.txt | cut -c1-220
{
for sh in "512 16" "512 32" "256 32" "128 32" "128 8"; do set -- $sh
  timeout 120 python tools/kbench.py gn_fwd --c $1 --hw $2 --iters 50 2>&1 | grep gn_fwd
  for res in 0 1; do timeout 120 python tools/kbench.py gn_bwd --c $1 --hw $2 --res $res --iters 50 2>&1 | grep "gn_bwd\[default\|gn_bwd\[three"; done
done
} > $O/kbench_gn_small.txt 2>&1; cat $O/kbench_gn_small.txt
for m in 1 0 1 0; do
  MAS_GN_SMALL=$m timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('gn_small=$m', d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('avg_launch_ms'), 'clock', r.get('sustained_clock_mhz'), 'MHz', r.get('package_power_w'), 'W; enc', d['encoder_stack']['fwd_ms'])"
done > $O/bench_ab.txt 2>&1; cat $O/bench_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-220
