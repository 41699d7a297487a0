#!/bin/bash
# round 4, session 4: the ticket-queue GroupNorm backward (gn_bwd_queue_kernel) -- parity, kbench vs three launches, plan sweep; round-4 parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gn_coop.py -x -q > $O/pytest_gn_queue.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gn_queue.txt
tail -5 $O/pytest_gn_queue.txt
{
for sh in "128 256" "128 128" "256 128" "256 64" "512 64" "512 32" "512 16" "128 64"; do set -- $sh
  for res in 0 1; do timeout 120 python tools/kbench.py gn_bwd --c $1 --hw $2 --res $res --iters 30 2>&1 | grep gn_bwd; done
done
} > $O/kbench_gn_bwd.txt 2>&1
cat $O/kbench_gn_bwd.txt
{
for T in 256 512; do for P in 1 2 3 4 8; do for L in 1 2 3; do
  [ $((T*P)) -gt 2048 ] && continue
  echo "== T=$T per_cu=$P lead=$L"
  MAS_GN_Q_THREADS=$T MAS_GN_Q_WGS_PER_CU=$P MAS_GN_Q_LEAD=$L timeout 120 python tools/kbench.py gn_bwd --c 128 --hw 256 --res 0 --three 0 --iters 30 2>&1 | grep gn_bwd
done; done; done
for U in 2 8 16; do echo "== units=$U"; MAS_GN_Q_UNITS=$U timeout 120 python tools/kbench.py gn_bwd --c 128 --hw 256 --three 0 --iters 30 2>&1 | grep gn_bwd; done
} > $O/sweep_gn_queue.txt 2>&1
cat $O/sweep_gn_queue.txt
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py -x -q -s > $O/pytest_parity_r4.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_parity_r4.txt
grep -v "^  warn\|Warning\|amdgpu.ids\|^  bf16\|^  fp32" $O/pytest_parity_r4.txt | tail -40 | cut -c1-230
