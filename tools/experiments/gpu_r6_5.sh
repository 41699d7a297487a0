#!/bin/bash
# round 6, call 5: XCD-banded tile walk of the wide kernel (MAS_CONV_XCD_BANDS=1 vs 0): parity tests, kbench A/B, FETCH_SIZE per launch, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_5; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --timeout 600 tests/test_gpu_wide.py tests/test_gpu_bn.py "tests/test_gpu_kernels.py::test_conv_full_size_properties" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
KB="timeout 100 python tools/kbench.py"
{
for rep in 1 2; do for b in 0 1; do
  echo "== MAS_CONV_XCD_BANDS=$b (rep $rep)"
  MAS_CONV_XCD_BANDS=$b $KB conv_fwd --n 32 --c 128 --hw 256 | tail -1
  MAS_CONV_XCD_BANDS=$b $KB conv_fwd --n 32 --c 128 --hw 256 --stats 1 --res 1 | tail -1
  MAS_CONV_XCD_BANDS=$b $KB conv_fwd --n 32 --c 256 --hw 64 | tail -1
  MAS_CONV_XCD_BANDS=$b $KB conv_fwd --n 32 --c 512 --hw 32 | tail -1
done; done
for b in 0 1; do echo "== FETCH_SIZE / WRITE_SIZE, MAS_CONV_XCD_BANDS=$b"; MAS_CONV_XCD_BANDS=$b bash tools/pmc_kernel.sh "FETCH_SIZE" conv_fwd --n 32 --c 128 --hw 256 | grep dispatches
  MAS_CONV_XCD_BANDS=$b bash tools/pmc_kernel.sh "TCC_HIT_sum TCC_MISS_sum" conv_fwd --n 32 --c 128 --hw 256 | grep dispatches; done
} > $O/kbench.txt 2>&1; cat $O/kbench.txt
for rep in 1 2; do for b in 0 1; do
  MAS_CONV_XCD_BANDS=$b timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bands=$b', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done | tee $O/step_ab.txt
