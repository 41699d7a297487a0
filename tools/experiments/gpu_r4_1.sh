#!/bin/bash
# round 4, session 1: the one-launch GroupNorm backward -- parity, kbench against the three-launch path, plan sweep, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gn_coop.py -x -q > $O/pytest_gn_coop.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gn_coop.txt
tail -5 $O/pytest_gn_coop.txt
{
for sh in "128 256" "128 128" "256 128" "256 64" "512 64" "512 32" "512 16" "128 64"; do set -- $sh
  for res in 0 1; do timeout 120 python tools/kbench.py gn_bwd --c $1 --hw $2 --res $res --iters 30 2>&1 | grep gn_bwd; done
done
} > $O/kbench_gn_bwd.txt 2>&1
cat $O/kbench_gn_bwd.txt
{
for T in 256 512 1024; do for P in 1 2 4; do for D in 1 2 3; do
  [ $((T*P)) -gt 1024 ] && continue
  echo "== T=$T per_cu=$P depth=$D"
  MAS_GN_COOP_THREADS=$T MAS_GN_COOP_WGS_PER_CU=$P MAS_GN_COOP_DEPTH=$D timeout 120 python tools/kbench.py gn_bwd --c 128 --hw 256 --res 0 --three 0 --iters 30 2>&1 | grep gn_bwd
done; done; done
for U in 2 8; do echo "== units=$U"; MAS_GN_COOP_UNITS=$U timeout 120 python tools/kbench.py gn_bwd --c 128 --hw 256 --three 0 --iters 30 2>&1 | grep gn_bwd; done
} > $O/sweep_gn_coop.txt 2>&1
cat $O/sweep_gn_coop.txt
for m in 1 0 1 0; do
  MAS_GN_COOP=$m timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('coop=$m', d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'dominant', d['roofline'].get('avg_launch_ms'))" 
done > $O/bench_ab.txt 2>&1
cat $O/bench_ab.txt
